"""Host logic: the circuit compiler and the noise-free integer circuit (no GPU)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def tiny():
    from dctfhe import compile as cc, models, params as P
    rng = np.random.default_rng(0)
    calib = rng.normal(0, 1, (48, 4, 6, 6))
    return cc.compile_model(models.tiny_resnet_q(), calib, param_set=P.test_params()), calib


def test_compile_is_deterministic(tiny):
    from dctfhe import compile as cc, models, params as P
    c, calib = tiny
    c2 = cc.compile_model(models.tiny_resnet_q(), calib, param_set=P.test_params())
    assert c.blob == c2.blob


def test_blob_round_trips_through_oracle_parser(tiny):
    from oracle import circuit_ref
    c, _ = tiny
    parsed = circuit_ref.parse_blob(c.blob)
    assert len(parsed["ops"]) == len(c.ops) and len(parsed["tensors"]) == len(c.tensors)
    for po, o in zip(parsed["ops"], c.ops):
        assert po["type"] == o.type and po["dst"] == o.dst and list(po["ip"][:7]) == [int(x) for x in o.ip[:7]]


def test_integer_circuit_reproduces_compile_time_values(tiny):
    """the oracle interpreter, fed the calibration inputs, lands on the values the compiler saw"""
    from dctfhe import compile as cc
    from oracle import circuit_ref
    c, calib = tiny
    q = cc.act_quant(calib, c.in_scale, True, c.in_bits)
    ph = (q.astype(np.int64).astype(np.uint64) << np.uint64(c.e_in)).reshape(q.shape[0], -1)
    out, overflow = circuit_ref.run_clear(c.blob, ph)
    assert not overflow
    vals = (out + (np.uint64(1) << np.uint64(c.e_out - 1))).view(np.int64) >> np.int64(c.e_out)
    assert vals.min() >= -8 and vals.max() <= 7 and len(np.unique(vals)) > 3


def test_rounding_semantics_round_half_up():
    from dctfhe import compile as cc
    m = np.arange(-64, 60)
    idx = cc.lut_index(m, 7, 1, True)
    assert np.array_equal(idx, (m + 64 + 1) >> 1)
    assert np.array_equal(cc.lut_centers(7, 1, 6, True), np.arange(64) * 2 - 64)
    assert cc._acc_precision(-100, 100, 6, 0.0) == (8, 2, True)
    assert cc._acc_precision(0, 15, 64, 0.0) == (4, 0, False)
    assert cc._acc_precision(-16, 15, 64, 0.0) == (5, 0, True)


def test_encodings_and_tiers_resnet20():
    from dctfhe import compile as cc, models
    from dctfhe.synthetic import synthetic_dct_batch
    c = cc.compile_model(models.ResNet20QAT(4, 24, 16), synthetic_dct_batch(24, seed=7))
    luts = [o for o in c.ops if o.type == cc.OP_LUT]
    sites = [o for o in luts if "(refresh)" not in o.note]
    assert len(sites) == 1 + 9 * 4 + 1          # stem + 4 sites per block + pool
    assert sum(c.tensors[o.src0].C * c.tensors[o.src0].H * c.tensors[o.src0].W for o in sites) == 380992
    # conv-feeding 5- and 6-bit sites are split into a coarse look-up and a small-ring refresh (stem + two per block)
    assert len(luts) - len(sites) == 19 and all(o.r == 0 and o.w <= 4 for o in luts if "(refresh)" in o.note)
    assert all(o.w <= 6 for o in luts) and c.max_bit_width <= 16
    for o in luts:
        assert o.ip[3] >= 0 and c.param_set.tiers[o.ip[4]].logN - 1 >= o.w
    assert c.expected_failures_per_image < 1e-3
    assert "round_lut" in c.report()


def test_approximate_rounding_and_p_error_policy(tiny):
    """approximate rounding drops every one-bit step (flag ip[9] of the blob record) and leaves the integer
    semantics alone; tier_policy "p_error" switches to the cheaper catalogue with every site within the budget."""
    from dctfhe import compile as cc, models, params as P
    from oracle import circuit_ref
    rng = np.random.default_rng(0)
    calib = rng.normal(0, 1, (32, 4, 6, 6))
    exact = cc.compile_model(models.tiny_resnet_q(), calib, param_set=P.test_params())
    approx = cc.compile_model(models.tiny_resnet_q(), calib, param_set=P.test_params(), rounding_method="approximate")
    luts = [o for o in approx.ops if o.type == cc.OP_LUT]
    assert any(o.r > 0 for o in luts) and all(o.ip[9] == (1 if o.r > 0 else 0) for o in luts)
    assert all(o.ip[9] == 0 for o in exact.ops if o.type == cc.OP_LUT)
    assert set(approx.pbs_counts()) == {"t"} and set(exact.pbs_counts()) == {"t", "b"}
    assert approx.expected_boundary_flips_per_image >= 0.0 and exact.expected_boundary_flips_per_image == 0.0
    q = cc.act_quant(calib[:3], exact.in_scale, True, exact.in_bits).astype(np.int64)
    ph = (q.astype(np.uint64) << np.uint64(exact.e_in)).reshape(3, -1)
    assert np.array_equal(circuit_ref.run_clear(exact.blob, ph)[0], circuit_ref.run_clear(approx.blob, ph)[0])
    fast = cc.compile_model(models.tiny_resnet_q(), calib, p_error=0.01, tier_policy="p_error", rounding_method="approximate")
    assert {t.name for t in fast.param_set.tiers} >= {"F6", "F5"}
    assert max(o.pfail for o in fast.ops if o.type == cc.OP_LUT) <= 0.01
    with pytest.raises(ValueError):
        cc.compile_model(models.tiny_resnet_q(), calib, p_error=0.5, tier_policy="p_error")


@pytest.mark.parametrize("name,in_ch,img,front", [("ResNet20QAT", 24, 16, "dct4"), ("ResNet20QAT", 3, 32, "rgb"), ("ResNet18QAT", 3, 32, "rgb"),
                                                  ("ResNet18QAT", 48, 112, "dct8")])
def test_exact_catalogue_keeps_every_site_in_budget(name, in_ch, img, front):
    """the default catalogue (n = 832 / 584, effective-dimension key switch) keeps every look-up site of the benchmark
    topologies under 1e-12 per element -- including ResNet-18's signed stem output, which is refreshed like the others
    (config #5 with four calibration images to keep the test short)"""
    from dctfhe import compile as cc, frontend, models, synthetic
    if front == "dct4":
        tf = frontend.dct_eval_transform(filter_size=4, image_size_dct=img, channels=in_ch)
    elif front == "dct8":
        tf = frontend.dct_eval_transform(filter_size=8, image_size_dct=img, channels=in_ch)
    else:
        tf = frontend.rgb_eval_transform(img)
    x = np.stack([tf(im) for im in synthetic.synthetic_images(24 if img <= 32 else 4, 7, size=64)]).astype(np.float32)
    model = getattr(models, name)(bit_width=4, in_channels=in_ch, img_size=img, seed=0)
    c = cc.compile_model(model, x, rounding_threshold_bits=6, n_bits=5)
    assert c.worst_site_failure <= 1e-12, c.worst_site_failure
    assert "T6" not in c.pbs_counts() and "T5" not in c.pbs_counts()        # every wide conv-feeding table is split


def _variant(changes):
    """default catalogue with some tiers replaced (noise levels re-derived from the new key lengths)"""
    import dataclasses
    from dctfhe import params as P
    ps = P.default_params()
    tiers = list(ps.tiers)
    for name, ch in changes.items():
        i = [t.name for t in tiers].index(name)
        tiers[i] = dataclasses.replace(tiers[i], lwe_sigma=0.0, **ch)
    return dataclasses.replace(ps, tiers=tiers)


def test_rounding_chain_hands_over_b_ba_ba2_in_order():
    """with the noisier base-8 key switch of the one-bit tiers (the catalogue before the base-4 gadgets) a rounding chain needs all three
    bit tiers: B first, then Ba, then Ba2 -- never back -- and the blob carries both hand-over points (ip[8], ip[11])"""
    from dctfhe import compile as cc, models
    from dctfhe.synthetic import synthetic_dct_batch
    old_bits = dict(n=584, lk=5, betak=3)
    ps = _variant({"B": old_bits, "Ba": old_bits, "Ba2": old_bits})
    c = cc.compile_model(models.ResNet20QAT(4, 24, 16), synthetic_dct_batch(24, seed=7), param_set=ps)
    counts = c.pbs_counts()
    assert counts["B"] > 0 and counts["Ba"] > 0 and counts["Ba2"] > counts["Ba"]
    names = [t.name for t in ps.tiers]
    seen_three = False
    for o in c.ops:
        if o.type != cc.OP_LUT or o.r == 0:
            continue
        chain = [names[cc.step_tier(o, i)] for i in range(o.r)]
        order = [{"B": 0, "Ba": 1, "Ba2": 2}[x] for x in chain]
        assert order == sorted(order), chain
        seen_three |= len(set(chain)) == 3
        if o.ip[11] >= 0:
            assert names[o.ip[11] >> 8] == "Ba2" and o.ip[8] <= (o.ip[11] & 255) < o.r
    assert seen_three
    assert c.worst_site_failure < 1e-12
    # the shipped catalogue: Ba2 takes every step Ba would
    d = cc.compile_model(models.ResNet20QAT(4, 24, 16), synthetic_dct_batch(24, seed=7)).pbs_counts()
    assert d.get("Ba", 0) == 0 and d["Ba2"] > 0 and d["B"] > 0


def test_table_tier_falls_back_when_a_site_leaves_the_budget():
    """a refresh tier that is too noisy for some site (here: T4r2 with 15-bit digits) sends the whole circuit to its quieter fallback T4r"""
    from dctfhe import compile as cc, models
    from dctfhe.synthetic import synthetic_dct_batch
    c = cc.compile_model(models.ResNet20QAT(4, 24, 16), synthetic_dct_batch(24, seed=7), param_set=_variant({"T4r2": dict(beta=15)}))
    counts = c.pbs_counts()
    assert counts.get("T4r2", 0) == 0 and counts["T4r"] > 0 and c.worst_site_failure < 1e-12
    assert cc.compile_model(models.ResNet20QAT(4, 24, 16), synthetic_dct_batch(24, seed=7)).pbs_counts().get("T4r", 0) == 0
