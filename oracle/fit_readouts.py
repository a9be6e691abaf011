"""Fitted read-outs for the synthetic networks -- TEST INFRASTRUCTURE ONLY (uses the torch reference networks and the CPU oracle).

Why (round-5 review, "missing" 5 / "next" 1c): under seeded random weights -- plain or the discriminating set of disc_weights.py -- the
refiner's deltas and the scorer's maximum have NOTHING to do with the pose error, so "Register returned pose X" cannot be compared with the
scene's ground truth and an 8-bit winner 140 degrees away from the f16 winner means nothing either way
(reference call sites: detection_6d_foundationpose/src/foundationpose.cpp:206-228, :432-446; the reference's own acceptance is visual,
simple_tests/src/test_foundationpose.cpp:48-104).

What: the seeded trunk and transformer layers stay as they are (seed 9 + tests/golden/disc_calib_seed9.npz).  Only the three LINEAR
read-outs are fitted, by ridge regression on pooled features of oracle crops of the synthetic TRAINING scenes
(synthetic.calibration_scenes: seeds 1000.., never the held-out scenes 5000.. nor the default scene):
  refiner trans_head.1 / rot_head.1 (Linear 512 -> 3 on the token mean; it commutes with the mean): targets = the deltas that
      RefinePostProcess (foundationpose.cpp:360-406) would need to land on the ground-truth pose -- (t_gt - t_hyp) / (diameter / 2) and
      atanh(axis-angle(R_hyp R_gt^T) / REFINE_ROT_NORMALIZER) -- for hypotheses within reach (sampler hypotheses near the truth + random
      perturbations of it);
  scorer linear (Linear 512 -> 1 behind att_cross): target = -(translation error / 20 mm + rotation error / 20 deg), clipped, on batches
      of sampler hypotheses refined by the fitted refiner (the batch composition att_cross sees in a Register).
The record is the disc record plus those three layers: `weights.make_synthetic_state(kind, 9, record)` applies it with numpy only.

RESULT (round 6, --scenes 16 = 32 training scenes, lambda by leave-SCENES-out validation; EXPERIMENTS.md R6.2): **the read-outs do not
generalise to scenes they were not fitted on** -- held-out-scene R^2: translation z 0.88, translation x / y -1.1, rotation -0.2 ... -0.07, score
-0.01.  z is carried almost linearly by the observed crop's xyz channels; x / y drown in the crop's background pixels, rotation and the score
need A-versus-B comparisons that seeded random features averaged over 400 tokens do not expose linearly.  So NO fixture is committed and NO test
uses this file: a synthetic end-to-end answer that means something needs trained weights.  Kept as the record of the attempt (and as the
generator, should a better feature basis turn up).

   python -m oracle.fit_readouts [--scenes 16] [--out /tmp/fit_calib_seed9.npz]      (~4 min on 8 cores)
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

from foundationpose_cpp_amd import synthetic as syn
from foundationpose_cpp_amd import weights as W

from . import fp_oracle as fo
from . import nets_torch as NT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 9
ROT_NORM = 0.349065850398865   # REFINE_ROT_NORMALIZER (foundationpose.cpp:82)


def log_so3(R):
    """rotation matrix -> axis-angle vector"""
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    ang = np.arccos(c)
    if ang < 1e-9:
        return np.zeros(3)
    ax = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(ang))
    return ax * ang


def needed_outputs(hyp, gt, diameter):
    """(trans[3], rot[3], rotation error in rad) that fpo_refine_post_process needs to map hyp onto gt: t' = t + trans * d/2, R' = R_delta^T R_hyp
    with R_delta = exp(tanh(rot) * ROT_NORM)  =>  R_delta = R_hyp R_gt^T"""
    trans = (gt[:3, 3].astype(np.float64) - hyp[:3, 3]) / (diameter / 2)
    v = log_so3(hyp[:3, :3].astype(np.float64) @ gt[:3, :3].astype(np.float64).T)
    rot = np.arctanh(np.clip(v / ROT_NORM, -0.95, 0.95))
    return trans, rot, float(np.linalg.norm(v))


def pose_errors(poses, gt):
    dt = np.linalg.norm(poses[:, :3, 3] - gt[:3, 3], axis=1)
    dR = np.einsum("nij,kj->nik", poses[:, :3, :3].astype(np.float64), gt[:3, :3].astype(np.float64))
    return dt, np.arccos(np.clip((np.trace(dR, axis1=1, axis2=2) - 1) / 2, -1, 1))


def perturbed(gt, rng, deg_lo, deg_hi, mm_lo, mm_hi):
    return syn.perturb_pose(gt, deg=float(rng.uniform(deg_lo, deg_hi)), trans=float(rng.uniform(mm_lo, mm_hi)) * 1e-3, seed=int(rng.integers(1 << 30)))


def blobs(om, mesh, scene, poses, ratio):
    p16 = syn.to_colmajor(poses)
    a = fo.render(om, p16, scene.K, scene.depth.shape, ratio)
    b = fo.crop(scene.rgb, scene.depth, scene.K, p16, ratio, mesh.diameter)
    return torch.from_numpy(a), torch.from_numpy(b)


def batched(fn, a, b, bs=12):
    outs = []
    with torch.no_grad():
        for i in range(0, len(a), bs):
            outs.append(fn(a[i:i + bs], b[i:i + bs]))
    if isinstance(outs[0], tuple):
        return tuple(torch.cat([o[k] for o in outs]) for k in range(len(outs[0])))
    return torch.cat(outs)


def ridge(X, Y, groups, lams=(1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 3e-2, 1e-1, 3e-1), weights=None):
    """centred ridge regression, lambda (relative to the mean feature variance) by leave-scenes-out validation -> (W [out,512], b [out], report)"""
    X = X.astype(np.float64); Y = Y.astype(np.float64)
    wts = np.ones(len(X)) if weights is None else weights.astype(np.float64)

    def solve(Xt, Yt, wt, lam):
        mu, my = np.average(Xt, 0, weights=wt), np.average(Yt, 0, weights=wt)
        Xc, Yc = (Xt - mu) * np.sqrt(wt)[:, None], (Yt - my) * np.sqrt(wt)[:, None]
        scale = (Xc ** 2).sum() / Xc.shape[1]
        Wm = np.linalg.solve(Xc.T @ Xc + lam * scale * np.eye(Xc.shape[1]), Xc.T @ Yc).T
        return Wm, my - Wm @ mu

    ug = np.unique(groups)
    folds = [ug[i::4] for i in range(4)]
    best = None
    for lam in lams:
        sse, sst = 0.0, 0.0
        for f in folds:
            te = np.isin(groups, f)
            Wm, bm = solve(X[~te], Y[~te], wts[~te], lam)
            pred = X[te] @ Wm.T + bm
            sse += (wts[te, None] * (pred - Y[te]) ** 2).sum(0)
            sst += (wts[te, None] * (Y[te] - np.average(Y[~te], 0, weights=wts[~te])) ** 2).sum(0)
        r2 = 1 - sse / sst
        if best is None or r2.mean() > best[1].mean():
            best = (lam, r2)
    Wm, bm = solve(X, Y, wts, best[0])
    return Wm.astype(np.float32), bm.astype(np.float32), dict(lam=best[0], r2_heldout_scenes=[round(float(x), 3) for x in best[1]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=16, help="training scenes per mesh (textured + untextured)")
    ap.add_argument("--out", default=os.path.join("/tmp", f"fit_calib_seed{SEED}.npz"))
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    disc = W.load_calibration(os.path.join(ROOT, "tests", "golden", f"disc_calib_seed{SEED}.npz"))
    refiner = NT.build("refiner", W.make_synthetic_state("refiner", SEED, disc))
    scorer = NT.build("scorer", W.make_synthetic_state("scorer", SEED, disc))
    rng = np.random.default_rng(4242)
    t0 = time.time()

    # ---- refiner: pooled tokens of both heads for hypotheses within reach of one refinement step
    Xt, Xr, Yt, Yr, G, ROK = [], [], [], [], [], []
    train = []
    for textured in (True, False):
        mesh = syn.make_mesh(textured=textured)
        om = fo.OracleMesh(mesh)
        for si, scene in enumerate(syn.calibration_scenes(mesh, args.scenes)):
            hyps = syn.from_colmajor(fo.get_hyp_poses(scene.depth, scene.mask, scene.K))
            _, rerr = pose_errors(hyps, scene.gt_pose)
            order = np.argsort(rerr)
            poses = [hyps[i] for i in order[:8]] + [hyps[i] for i in rng.choice(order[8:], 4, replace=False)]
            poses += [perturbed(scene.gt_pose, rng, 0, 10, 0, 8) for _ in range(10)] + [perturbed(scene.gt_pose, rng, 10, 30, 5, 30) for _ in range(10)]
            # sampler-like translation (guess from the depth) with a rotation within reach
            for _ in range(6):
                p = perturbed(scene.gt_pose, rng, 5, 35, 0, 1)
                p[:3, 3] = hyps[0][:3, 3]
                poses.append(p)
            poses = np.stack(poses).astype(np.float32)
            a, b = blobs(om, mesh, scene, poses, 1.2)

            def pooled(aa, bb):
                f = NT._trunk(refiner, aa, bb)
                return refiner.trans_head[0](f).mean(1), refiner.rot_head[0](f).mean(1)
            mt, mr = batched(pooled, a, b)
            for k, p in enumerate(poses):
                tr, ro, ang = needed_outputs(p, scene.gt_pose, mesh.diameter)
                Yt.append(tr); Yr.append(ro); ROK.append(ang < np.deg2rad(40))
            Xt.append(mt.numpy()); Xr.append(mr.numpy()); G += [len(train)] * len(poses)
            train.append((textured, si))
            print(f"refiner features: scene {len(train)} ({'textured' if textured else 'untextured'} {si}), {len(poses)} hypotheses, {time.time() - t0:.0f} s", flush=True)
    Xt, Xr, Yt, Yr, G, ROK = np.concatenate(Xt), np.concatenate(Xr), np.array(Yt), np.array(Yr), np.array(G), np.array(ROK)
    TOK = np.linalg.norm(Yt, axis=1) < 0.8     # translations within ~4 cm of the truth (diameter / 2 = 0.095 m)
    Wt, bt, rep_t = ridge(Xt[TOK], Yt[TOK], G[TOK])
    Wr, br, rep_r = ridge(Xr[ROK], Yr[ROK], G[ROK])
    print("trans head:", rep_t, " rot head:", rep_r, flush=True)
    rec = dict(disc)
    rec["refiner/trans_head.1.weight"], rec["refiner/trans_head.1.bias"] = Wt, bt
    rec["refiner/rot_head.1.weight"], rec["refiner/rot_head.1.bias"] = Wr, br
    refiner = NT.build("refiner", W.make_synthetic_state("refiner", SEED, rec))

    # ---- scorer: a third of every training scene's sampler hypotheses, refined ONCE by the fitted refiner (what a Register scores)
    Xs, Ys, Gs = [], [], []
    gi = 0
    for textured in (True, False):
        mesh = syn.make_mesh(textured=textured)
        om = fo.OracleMesh(mesh)
        for si, scene in enumerate(syn.calibration_scenes(mesh, args.scenes)):
            hyps16 = fo.get_hyp_poses(scene.depth, scene.mask, scene.K)
            _, rerr = pose_errors(syn.from_colmajor(hyps16), scene.gt_pose)
            sel = np.unique(np.concatenate([np.arange(si % 3, 252, 3), np.argsort(rerr)[:6]]))
            p16 = hyps16[sel]
            a, b = blobs(om, mesh, scene, syn.from_colmajor(p16), 1.2)
            t, r = batched(lambda aa, bb: refiner(aa, bb), a, b)
            refined = syn.from_colmajor(fo.refine_post_process(p16, t.numpy(), r.numpy(), mesh.diameter))
            a, b = blobs(om, mesh, scene, refined, 1.1)
            feats = batched(lambda aa, bb: scorer.extract_feat(aa, bb), a, b)
            with torch.no_grad():
                x, _ = scorer.att_cross(feats[None], feats[None], feats[None], need_weights=False)
            dt, dr = pose_errors(refined, scene.gt_pose)
            Xs.append(x[0].numpy()); Ys.append(-np.minimum(dt * 1e3 / 20.0 + np.degrees(dr) / 20.0, 6.0)); Gs += [gi] * len(sel)
            gi += 1
            print(f"scorer features: scene {gi}, {len(sel)} hypotheses, refined errors: best {dt.min() * 1e3:.1f} mm / {np.degrees(dr.min()):.1f} deg, {time.time() - t0:.0f} s", flush=True)
    Xs, Ys, Gs = np.concatenate(Xs), np.concatenate(Ys)[:, None], np.array(Gs)
    # the few hypotheses near the truth decide the winner: weigh them up
    wts = 1.0 + 4.0 * (Ys[:, 0] > -2.0)
    Ws, bs, rep_s = ridge(Xs, Ys, Gs, weights=wts)
    print("score read-out:", rep_s, flush=True)
    rec["scorer/linear.weight"], rec["scorer/linear.bias"] = Ws, bs
    np.savez_compressed(args.out, **rec)
    print(f"wrote {args.out} ({os.path.getsize(args.out) / 1024:.0f} KB, {time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
