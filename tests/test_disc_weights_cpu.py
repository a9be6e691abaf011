"""The DISCRIMINATING synthetic weight set (oracle/disc_weights.py -> tests/golden/disc_calib_seed9.npz): the recipe reproduces
the committed record, the record applies with numpy only, and under it the oracle networks really tell hypotheses apart
(round-2 review, weak #1: under the plain draws the 252 scores agree to 3e-5, so no end-to-end check could see wiring errors)."""
import numpy as np
import torch

from foundationpose_cpp_amd import synthetic as syn, weights as W
from oracle import disc_weights as DW
from oracle import fp_oracle as fo
from oracle import nets_torch as NT

from conftest import DISC_SEED


def test_recipe_reproduces_the_committed_record(disc_cal):
    rec = {**DW.make_calibration("refiner", DISC_SEED), **DW.make_calibration("scorer", DISC_SEED)}
    assert sorted(rec) == sorted(disc_cal)
    for k, v in rec.items():
        # torch CPU reductions differ in the last bits between hosts / thread counts
        np.testing.assert_allclose(np.asarray(v), disc_cal[k], rtol=2e-3, atol=2e-4 * float(np.abs(disc_cal[k]).max()), err_msg=k)


def test_record_applies_with_numpy_only_and_keeps_the_architecture(disc_cal):
    for kind in ("refiner", "scorer"):
        plain = W.make_synthetic_state(kind, DISC_SEED)
        disc = W.make_synthetic_state(kind, DISC_SEED, disc_cal)
        assert sorted(plain) == sorted(disc)
        changed = [k for k in plain if not np.array_equal(plain[k], disc[k])]
        # only heads change: the convolution trunk keeps the plain (well-conditioned) draws
        assert changed and all(not k.startswith("encode") for k in changed), changed
        NT.build(kind, disc)     # strict load
        folded = W.fold_batchnorm(disc)
        assert all(np.isfinite(v).all() for v in folded.values())


def test_oracle_networks_discriminate(disc_nets, syn_mesh, syn_scene):
    """42 sampler hypotheses (every 6th): refiner outputs vary by >= 30 % of their magnitude, score spread >= 0.5, unique maximum"""
    om = fo.OracleMesh(syn_mesh)
    poses = fo.get_hyp_poses(syn_scene.depth, syn_scene.mask, syn_scene.K)[::6]
    a = fo.render(om, poses, syn_scene.K, syn_scene.depth.shape, 1.2)
    b = fo.crop(syn_scene.rgb, syn_scene.depth, syn_scene.K, poses, 1.2, syn_mesh.diameter)
    with torch.no_grad():
        t, r = (x.numpy() for x in disc_nets[2](torch.from_numpy(a), torch.from_numpy(b)))
    for y in (t, r):
        assert (y.std(0) >= 0.3 * np.sqrt((y ** 2).mean(0))).all(), (y.std(0), np.sqrt((y ** 2).mean(0)))
    refined = fo.refine_post_process(poses, t, r, syn_mesh.diameter)
    a = fo.render(om, refined, syn_scene.K, syn_scene.depth.shape, 1.1)
    b = fo.crop(syn_scene.rgb, syn_scene.depth, syn_scene.K, refined, 1.1, syn_mesh.diameter)
    with torch.no_grad():
        s = disc_nets[3](torch.from_numpy(a), torch.from_numpy(b)).numpy()
    o = np.sort(s)[::-1]
    assert s.std() >= 0.5 and o[0] - o[1] >= 1e-2, (s.std(), o[:3])
    # the plain draws of the same seed, for the record: three orders of magnitude less spread
    plain = NT.build("scorer", W.make_synthetic_state("scorer", DISC_SEED))
    with torch.no_grad():
        sp = plain(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    assert sp.std() < 1e-3 * s.std()
