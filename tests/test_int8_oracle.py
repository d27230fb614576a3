"""Pin the integer-exact int8 oracle (oracle/int8_forward.py) on the CPU: known answers for its float epilogue, its two conv
back-ends against each other, its upsample + add against the Caffe-semantics deconvolution of the fp32 oracle, and the whole
int8 network against the fp32 oracle (quantisation noise only).  The GPU engine is then held BIT-exact to it (-m gpu)."""
import os

import numpy as np
import pytest

from conftest import STEMS
from oracle import int8_forward as i8
from oracle.caffe_forward import _deconv2d_numpy
from oracle.retinaface_post import decode, iou_plus1, nms, preprocess_trt_identity


def _requant(acc, mult, bias, relu=True):
    acc = np.asarray(acc, np.int32).reshape(1, -1, 1)
    n = acc.shape[1]
    out = np.empty((1, n, 1), np.int8)
    for k in range(n):      # one channel per call so every entry can have its own (mult, bias)
        a = np.ascontiguousarray(acc[:, k:k + 1])
        o = i8.Int8Net._requant(a, np.array([mult[k]], np.float32), np.array([bias[k]], np.float32), relu)
        out[0, k, 0] = o[0, 0, 0]
    return out.reshape(-1).tolist()


def test_requantisation_known_answers():
    """y = fmaf(acc, mult, bias) with ONE rounding, then round-half-even of the clamped value."""
    f = float.fromhex
    # ties go to the even integer; negative -> 0 (ReLU); saturation at 127; -127 floor without ReLU
    assert _requant([5, 7, 9, -3, 1000, 254], [0.5] * 6, [0.0] * 6) == [2, 4, 4, 0, 127, 127]
    assert _requant([-5, -7, -1000, 5], [0.5] * 4, [0.0] * 4, relu=False) == [-2, -4, -127, 2]
    # vectors where fl(fl(acc * mult) + bias) and fmaf(acc, mult, bias) land on different sides of k + 0.5 (found by search):
    # the oracle and the kernels (v_fma_f32) take the fused value
    cases = [(95164, f("0x1.391236p-7"), f("-0x1.a7db4ep+9"), 61, 62), (129121, f("0x1.472a02p-10"), f("-0x1.7a96aap+6"), 66, 67),
             (180870, f("0x1.b88268p-9"), f("-0x1.232f8ep+9"), 25, 26), (150071, f("0x1.ed08d6p-8"), f("-0x1.fac014p+9"), 115, 116)]
    for acc, m, b, fused, unfused in cases:
        assert _requant([acc], [m], [b]) == [fused]
        two_step = np.float32(np.float32(np.float32(acc) * np.float32(m)) + np.float32(b))
        assert int(np.rint(two_step)) == unfused
    # depthwise outputs (round 6): ReLU'd quanta 0..255 stored minus 128 (relu = 2): ties to even, saturation at 255, negatives -> 0
    assert _requant([5, 7, -3, 1000, 509, 511, 256], [0.5] * 7, [0.0] * 7, relu=2) == [2 - 128, 4 - 128, -128, 127, 254 - 128, 127, 0]
    # int32 -> float conversion of the accumulator rounds to nearest even above 2**24 (15-bit depthwise taps reach 1.9e7)
    assert _requant([2 ** 24 + 1, 2 ** 24 + 3], [2.0 ** -18] * 2, [0.0] * 2) == [64, 64]


def test_integer_blend_equals_caffe_deconv_plus_eltwise(nets):
    """rfi8_upadd mode 1 (integers, round half to even on the total) == rint(Deconvolution(k4 s2 p1 g, bilinear weights) + Crop +
    Eltwise SUM) computed with the fp32 oracle's own transposed convolution on the same quanta; mode 0 (fp32 form) with unit
    scale ratios agrees with it bit for bit, and with non-unit ratios follows fmaf(lat, a_lat, blend * a_up)."""
    rng = np.random.default_rng(5)
    net = i8.Int8Net(nets["mnet25"])
    h, w, c = 10, 14, 64
    lat = rng.integers(0, 128, (h, w, c)).astype(np.int8)
    up = rng.integers(0, 128, (h // 2, w // 2, c)).astype(np.int8)
    k1 = np.array([0.25, 0.75, 0.75, 0.25], np.float32)
    wdec = np.tile(np.outer(k1, k1)[None, None], (c, 1, 1, 1)).astype(np.float32)          # [cin][cout/g = 1][4][4]
    dec = _deconv2d_numpy(up.transpose(2, 0, 1)[None].astype(np.float32), wdec, None, 2, 1, c)[0].transpose(1, 2, 0)
    want = np.minimum(np.rint(dec.astype(np.float64) + lat), 127).astype(np.int8)         # multiples of 1/16: exact, half to even
    out = np.empty_like(lat)
    i8.lib().rfi8_upadd(lat.ctypes.data, up.ctypes.data, h, w, c, 1.0, 1.0, 1, out.ctypes.data)
    assert np.array_equal(out, want)
    out0 = np.empty_like(lat)
    i8.lib().rfi8_upadd(lat.ctypes.data, up.ctypes.data, h, w, c, 1.0, 1.0, 0, out0.ctypes.data)
    assert np.array_equal(out0, want)
    a_lat, a_up = np.float32(0.7312), np.float32(1.318)
    i8.lib().rfi8_upadd(lat.ctypes.data, up.ctypes.data, h, w, c, float(a_lat), float(a_up), 0, out0.ctypes.data)
    real = lat.astype(np.float64) * float(a_lat) + dec.astype(np.float64) * float(a_up)
    assert np.abs(out0 - np.clip(real, -127, 127)).max() <= 0.5 + 1e-4
    assert net.per_channel and net.a_lat == [1.0, 1.0]


@pytest.mark.parametrize("stem", STEMS)
def test_int8_conv_backends_agree_exactly(nets, stem):
    """float64 BLAS convolution (exact: every partial sum is an integer below 2**53) vs the plain C loops, every activation of
    the network, on random quanta at a 64 x 96 net size."""
    rng = np.random.default_rng(11)
    x = rng.integers(0, 128, (32, 48, 16)).astype(np.int8)
    a = i8.Int8Net(nets[stem], backend="blas").forward_from("mobilenet0_relu2_fwd", x)
    b = i8.Int8Net(nets[stem], backend="c").forward_from("mobilenet0_relu2_fwd", x)
    assert set(a) == set(b) and len(a) > 40
    for k in a:
        if k == "__heads__":
            for hk in a[k]:
                assert np.array_equal(a[k][hk], b[k][hk]), hk
        else:
            assert a[k].dtype == np.int8 and np.array_equal(a[k], b[k]), k
    # later starting points continue identically (the engine's fused front end hands over at relu4)
    c = i8.Int8Net(nets[stem]).forward_from("mobilenet0_relu4_fwd", a["mobilenet0_relu4_fwd"])
    assert all(np.array_equal(c[k], a[k]) for k in c if k != "__heads__")


@pytest.mark.parametrize("stem", STEMS)
def test_int8_oracle_tracks_the_fp32_oracle(nets, oracles, stem):
    """The int8 definition is sane: fed the fp32 oracle's own first block output, it finds the same faces (IoU / anchor
    agreement at the level the engine's INT8_BAR documents) and every layer stays within a few percent of the range."""
    from retinaface_amd.frames import synth_frames
    q = i8.Int8Net(nets[stem])
    ious, same_anchor, faces = [], 0, 0
    for f in synth_frames(448, 448, 4, config=300, faces=[1, 3, 5]):
        blobs = oracles[stem].forward(preprocess_trt_identity(f, 448, 448), keep_all=True)
        x = q.quantise_blob("mobilenet0_relu2_fwd", blobs["mobilenet0_relu2_fwd"][0].transpose(1, 2, 0))
        acts = q.forward_from("mobilenet0_relu2_fwd", x)
        for n in ("mobilenet0_relu10_fwd", "mobilenet0_relu26_fwd", "rf_c1_aggr_relu", "rf_c2_det_concat_relu"):
            r = blobs[n][0].transpose(1, 2, 0)
            a = acts[n].astype(np.float32) * q.scale_of_blob[n]
            assert np.abs(a - np.minimum(r, a.max())).mean() <= 0.035 * max(1.0, float(np.abs(r).max())), n
        heads = {k: v[None] for k, v in acts["__heads__"].items()}
        got = nms(list(decode(heads, 448, 448, 0.5)), 0.4)
        ref = oracles[stem].detect(f, 0.5, 0.4, net_hw=(448, 448)).detections
        assert len(got) == len(ref)
        for r in ref:
            best = max(got, key=lambda g: iou_plus1(g.rect, r.rect))
            ious.append(iou_plus1(best.rect, r.rect))
            same_anchor += best.anchor_index == r.anchor_index
            faces += 1
    assert faces >= 4 and min(ious) >= 0.85 and same_anchor / faces >= 0.5, (min(ious), same_anchor, faces)


def test_per_channel_table_with_equal_channels_equals_the_per_tensor_table(nets):
    """The per-channel extension of the table format (`blob#c` lines) must reduce to the TensorRT per-tensor semantics when every
    channel carries the tensor's scale -- except for the FPN adds, where per-channel tables give the three tensors of an add one
    common scale (the largest) and blend in integers: with equal scales for all three the two blends are the same function, so the
    whole network must agree activation for activation."""
    import copy
    from conftest import ASSETS
    from oracle.caffe_io import read_int8_table
    net = copy.deepcopy(nets["mnet-deconv-0517"])
    net.int8_qweights = {}            # the per-tensor table is the reference's TensorRT cache (assets/mnet-deconv-0517.table.int8), rounded to nearest
    net.int8_scales = read_int8_table(os.path.join(ASSETS, "mnet-deconv-0517.table.int8"))
    base = dict(net.int8_scales)
    # force the three tensors of each add to one per-tensor scale, so that the per-tensor engine's ratios are exactly 1
    for group in (("rf_c3_lateral_relu", "rf_c2_lateral_relu", "_plus0"), ("rf_c2_aggr_relu", "rf_c1_red_conv_relu", "_plus1")):
        m = max(base[g] for g in group)
        for g in group:
            base[g] = m
    net.int8_scales = dict(base)
    per_tensor = i8.Int8Net(net)
    assert not per_tensor.per_channel and per_tensor.a_lat == [1.0, 1.0] and per_tensor.a_up == [1.0, 1.0]
    chans = {"_plus0": 64, "_plus1": 64}
    for i in range(13):
        c = i8.Int8Net.BLOCK_COUT[i]
        chans[f"mobilenet0_relu{2 * i + 2}_fwd"] = c
        chans[f"mobilenet0_relu{2 * i + 1}_fwd"] = 8 if i == 0 else i8.Int8Net.BLOCK_COUT[i - 1]
    for n in ("rf_c3_lateral_relu", "rf_c2_lateral_relu", "rf_c1_red_conv_relu", "rf_c2_aggr_relu", "rf_c1_aggr_relu"):
        chans[n] = 64
    for c in (3, 2, 1):
        chans[f"rf_c{c}_det_concat_relu"] = 64
        chans[f"rf_c{c}_det_context_conv1_relu"] = 16
        chans[f"rf_c{c}_det_context_conv3_1_relu"] = 16
    wide = dict(base)
    for n, c in chans.items():
        for k in range(c):
            wide[f"{n}#{k}"] = base[n]
    net2 = copy.deepcopy(net)
    net2.int8_scales = wide
    per_channel = i8.Int8Net(net2)
    assert per_channel.per_channel
    rng = np.random.default_rng(23)
    x = rng.integers(0, 128, (32, 48, 16)).astype(np.int8)
    a = per_tensor.forward_from("mobilenet0_relu2_fwd", x)
    b = per_channel.forward_from("mobilenet0_relu2_fwd", x)
    for k in a:
        if k == "__heads__":
            assert all(np.array_equal(a[k][h], b[k][h]) for h in a[k])
        else:
            assert np.array_equal(a[k], b[k]), k


def test_u8_mid_algebra_and_calibrated_weights(nets):
    """A pointwise conv whose input is a depthwise mid stored as q - 128: sum_k w_q (q - 128) with the bias fmaf(mult, 128 * sum_k w_q, bias)
    is the conv over the unsigned quanta (exact integers; one extra fp32 rounding in the bias), the mid's scale is the table's x fp32(127/255),
    and calibrated weights replace the rounding but not the grid."""
    rng = np.random.default_rng(3)
    w = rng.standard_normal((32, 1, 1, 16)).astype(np.float32) * 0.1
    b = rng.standard_normal(32).astype(np.float32)
    s_in = rng.uniform(0.01, 0.05, 16).astype(np.float32)
    s_out = rng.uniform(0.02, 0.06, 32).astype(np.float32)
    g = i8.QGemm(w, b, s_in, s_out, in_u8=True)
    plain = i8.QGemm(w, b, (s_in * i8.MID_U8).astype(np.float32), s_out)
    assert np.array_equal(g.wq, plain.wq) and np.array_equal(g.mult, plain.mult)
    q = rng.integers(0, 256, (5, 7, 16))
    acc_u = np.einsum("hwk,ok->hwo", q, plain.wq.reshape(32, 16))
    acc_s = np.einsum("hwk,ok->hwo", q - 128, g.wq.reshape(32, 16))
    y_u = acc_u * plain.mult.astype(np.float64) + plain.bias.astype(np.float64)
    y_s = acc_s * g.mult.astype(np.float64) + g.bias.astype(np.float64)
    assert np.abs(y_u - y_s).max() <= 4e-6 * max(1.0, float(np.abs(y_u).max()))          # the only difference: fp32 rounding of the folded bias
    # calibrated integers: taken as they are, same multipliers, bias shifted by bias_delta / s_out
    cq = np.clip(plain.wq.reshape(32, 16) + rng.integers(-1, 2, (32, 16)), -127, 127).astype(np.int8)
    db = rng.standard_normal(32).astype(np.float32) * 0.01
    c = i8.QGemm(w, b, s_in, s_out, calibrated=(cq, db))
    n = i8.QGemm(w, b, s_in, s_out)
    assert np.array_equal(c.wq.reshape(32, 16), cq) and np.array_equal(c.mult, n.mult)
    assert np.array_equal(c.bias, ((b + db).astype(np.float32) / s_out).astype(np.float32))
    # the shipped models carry calibrated weights for every fused dense conv, all inside the int8 grid, and they differ from plain rounding
    for stem in STEMS:
        net = nets[stem]
        assert len(net.int8_qweights) == 29
        q8 = i8.Int8Net(net)
        import copy
        rtn = copy.copy(net)
        rtn.int8_qweights = {}
        q0 = i8.Int8Net(rtn)
        moved = np.mean([np.mean(a.wq != b.wq) for a, b in zip(q8.pw[1:], q0.pw[1:])])
        assert 0.03 < moved < 0.4, moved
        assert all(np.abs(a.wq - b.wq).max() <= 3 for a, b in zip(q8.pw[1:], q0.pw[1:]))
