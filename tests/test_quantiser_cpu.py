"""The INT8 weight quantiser of fp_nn.hip (quantise_q8) on the host, no GPU: the properties DESIGN.md section 4.4 claims for it.

With round-to-nearest the rounding errors of a row are independent: per input channel their TAP SUM T_c has a standard deviation of ~0.87
steps, and the row's mean output error sum_c T_c mean(x_c) random-walks over the channels.  The error-feedback form flips the weights
nearest a half step so that |T_c| <= 0.5 and the sums against every calibration frame's channel means stay near zero -- while no weight
moves by more than one step and only weights within TAU = 0.2 of .5 move at all."""
import ctypes as C

import numpy as np

from foundationpose_cpp_amd import _lib


def _quantise(rows, s_in, m_int):
    L = _lib.test_lib()
    Cout, nt, Cin = rows.shape
    q = np.zeros(rows.shape, np.int8)
    sw = np.zeros(Cout, np.float32)
    tm = np.zeros((Cin, Cout), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
    J = 0 if m_int is None else m_int.shape[0]
    L.fpt_quantise_i8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.fpt_quantise_i8(p(rows), Cout, nt, Cin, p(s_in), p(m_int), J, p(q), p(sw), p(tm)) == 0
    return q, sw, tm


def test_error_feedback_rounding_properties():
    rng = np.random.default_rng(0)
    Cout, nt, Cin, J = 64, 9, 256, 8
    rows = (rng.standard_normal((Cout, nt, Cin)) / np.sqrt(nt * Cin)).astype(np.float32)
    rows[:, :, :8] *= 6.0                                        # a few heavy channels (a large weight on a large activation scale)
    s_in = np.exp(rng.normal(0, 0.7, Cin)).astype(np.float32)    # per-input-channel activation scales spread over ~1.5 decades
    m_bar = rng.uniform(10, 60, Cin)
    m_int = (m_bar[None] * (1 + 0.2 * rng.standard_normal((J, Cin)))).astype(np.float32)    # frames' mean integer activations: +-20 % around a family mean
    q_n, sw_n, tm_n = _quantise(rows, s_in, None)                # round to nearest, step = amax / 127
    q_e, sw_e, tm_e = _quantise(rows, s_in, m_int)
    wf = rows * s_in[None, None, :]
    # nearest: |q - v| <= 1/2 everywhere, step = amax / 127
    v_n = wf / sw_n[:, None, None]
    assert np.allclose(sw_n, np.abs(wf).reshape(Cout, -1).max(1) / 127, rtol=1e-6)
    assert np.abs(q_n - v_n).max() <= 0.5 + 1e-4
    # error feedback: the step never grows, clipped weights aside nobody moves by a full step, and only near-half fractions move
    assert (sw_e <= sw_n * (1 + 1e-6)).all() and (sw_e >= 0.5 * sw_n * (1 - 1e-6)).all()
    v_e = wf / sw_e[:, None, None]
    inside = np.abs(v_e) < 126
    d = (q_e - v_e)[inside]
    assert np.abs(d).max() < 1.0
    moved = np.abs(d) > 0.5 + 1e-4
    frac = (v_e - np.floor(v_e))[inside]
    assert (np.abs(frac[moved] - 0.5) < 0.2 + 1e-4).all()
    assert 0.01 < moved.mean() < 0.25                           # a minority of the weights carries the correction
    # the tap sums: reported T equals sum_taps (q - v), is within +-0.5 almost everywhere (nearest: sd ~0.87)
    T_e = (q_e - v_e).sum(1)                                    # [Cout, Cin]
    assert np.allclose(T_e, tm_e.T, atol=2e-3)
    T_n = (q_n - v_n).sum(1)
    assert np.allclose(T_n, tm_n.T, atol=2e-3)
    ok_rows = ~(np.abs(v_e) >= 126).any(1)                      # (channels of a row that hold a clipped weight carry its clipping error)
    share = np.mean(np.abs(T_e[ok_rows]) <= 0.5 + 1e-3)
    print(f"tap sums: nearest sd {T_n.std():.3f}, |T| <= .5 on {np.mean(np.abs(T_n) <= 0.5) * 100:.0f} % of the (row, channel) pairs; error feedback sd {T_e[ok_rows].std():.3f}, {share * 100:.0f} %")
    assert share > 0.70 and T_n.std() > 0.7 and T_e[ok_rows].std() < 0.55 * T_n.std(), (share, T_n.std(), T_e[ok_rows].std())
    # the mean error against the calibration frames AND against unseen frames of the family (other +-20 % deviations)
    unseen = m_bar[None] * (1 + 0.2 * rng.standard_normal((16, Cin)))
    for name, frames in (("calibration", m_int.astype(np.float64)), ("unseen", unseen)):
        e_n = np.abs(frames @ T_n.T * sw_n[None]).mean()        # |sum_c T_c m_c| in real units, mean over frames and rows
        e_e = np.abs(frames @ T_e.T * sw_e[None]).mean()
        print(f"{name} frames: mean |row mean error| nearest {e_n:.4g} -> error feedback {e_e:.4g} ({e_n / e_e:.1f}x)")
        assert e_e < (0.15 if name == "calibration" else 0.45) * e_n, (name, e_n, e_e)
    # the de-meaned part does not pay for it: rms rounding error per weight (in units of the NEAREST step) grows by < 25 %
    rms_n = np.sqrt(((q_n - v_n) ** 2).mean())
    rms_e = np.sqrt((((q_e - v_e) * (sw_e / sw_n)[:, None, None]) ** 2)[inside.reshape(q_e.shape)].mean())
    assert rms_e < 1.25 * rms_n, (rms_n, rms_e)


def test_quantiser_is_deterministic_and_handles_degenerate_rows():
    rng = np.random.default_rng(1)
    rows = rng.standard_normal((16, 9, 128)).astype(np.float32)
    rows[3] = 0.0                                               # an all-zero row
    s_in = np.ones(128, np.float32)
    m_int = rng.uniform(0, 50, (4, 128)).astype(np.float32)
    m_int[:, 5] = 0.0                                           # a dead channel
    a = _quantise(rows, s_in, m_int)
    b = _quantise(rows, s_in, m_int)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert (a[0][3] == 0).all() and a[1][3] == 1.0
    assert np.isfinite(a[2]).all()
