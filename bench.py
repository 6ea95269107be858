#!/usr/bin/env python
"""bench.py -- encrypted images/s of the DCT-CryptoNets hot path on N MI355X GPUs.

Metric (BASELINE.json): encrypted images/sec (+ s/image), ResNet-20 DCT 24x16^2 CIFAR-10 trunk.
A "step" is one pass of the homomorphic circuit (dctfhe_session_run: every conv, add, exact rounding,
key switch and bootstrap of the trunk) over one batch of synthetic encrypted images that is already
resident in HBM.  The batch shards by image across ranks (one process per GPU); there is no collective
on the data path -- the only exchange is one RCCL all_gather of the decrypted-side logits at the end.

Usage:  python bench.py --gpus N --steps K --warmup W [--batch-per-gpu B]
        (N > 1: launched by torch.distributed.run, one rank per GPU)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dct-cryptonets_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_SPEC_TFLOPS = 78.6          # MI355X vector FP64 (spec; the guide does not list it, so the live FMA probe is reported beside it)


def _dct_batch(n, seed):
    from dctfhe.synthetic import synthetic_dct_batch
    return synthetic_dct_batch(n, seed=seed)


def _rgb_batch(n, seed):
    from dctfhe import frontend, synthetic
    tf = frontend.rgb_eval_transform(32)
    return np.stack([tf(im) for im in synthetic.synthetic_images(n, seed)]).astype(np.float32)


def _dct8_112_batch(n, seed):
    """config #5: 8x8 JPEG-domain DCT, 48 channels, 112x112: Resize(1030) -> CenterCrop(896) -> 8x8 DCT"""
    from dctfhe import frontend, synthetic
    tf = frontend.dct_eval_transform(filter_size=8, image_size_dct=112, channels=48)
    return np.stack([tf(im) for im in synthetic.synthetic_images(n, seed, size=96)]).astype(np.float32)


# name -> (model factory name, in_channels, img_size, input batch maker, description)
CONFIGS = {
    "r20_24_16": ("ResNet20QAT", 24, 16, _dct_batch, "ResNet-20 24x16^2 DCT CIFAR-10 trunk (BASELINE config #2 shape)"),
    "r20_3_32": ("ResNet20QAT", 3, 32, _rgb_batch, "ResNet-20 3x32^2 RGB CIFAR-10 trunk (BASELINE config #3)"),
    "r18_3_32": ("ResNet18QAT", 3, 32, _rgb_batch, "ResNet-18 3x32^2 RGB CIFAR-10 trunk (BASELINE config #4 shape)"),
    "r18_48_112": ("ResNet18QAT", 48, 112, _dct8_112_batch, "ResNet-18 48x112^2 DCT ImageNet trunk (BASELINE config #5 shape; 16 calibration images)"),
}


def measured_hbm_traffic(kernel_tag, cts_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (tools/pmc_summary.py; FETCH_SIZE and
    WRITE_SIZE need their own rocprofv3 passes, so they cannot be collected inside this run).  Scaled by ciphertexts per
    launch; None when the summary has no entry for this kernel."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_hbm.json")
    if not os.path.exists(path):
        return None
    prof = json.load(open(path))
    for k, v in prof.items():
        if kernel_tag in k.replace(" ", "") and v.get("launches"):
            per_ct = v["hbm_bytes_per_launch"] / v.get("cts_per_launch", cts_per_launch)
            return per_ct * cts_per_launch
    return None


def cpu_baseline(qm, stats, n_prime=16):
    """Times the CPU oracle (oracle/tfhe_ref.c, the C twin) on a bounded sample and extrapolates to images/s.
    Sample: per tier, `threads` ciphertexts through n'=16 blind-rotate iterations and a key switch onto n'+1
    columns -- both costs are exactly linear in n, so they are scaled by n/n' -- plus one 3x3 ciphertext conv."""
    # the GPU box shares its host cores: keep to the one-GPU CPU share (16), and say how many were used
    from oracle import ref_loader as R
    R.build()
    R.lib().ref_set_num_threads(min(16, os.cpu_count() or 1))
    ps = qm.compiled.param_set
    threads = R.lib().ref_num_threads()
    D = ps.D
    S = R.gen_binary_key(1, D)
    total_s = 0.0
    detail = {}
    for ti, t in enumerate(ps.tiers):
        cnt = int(stats.pbs_count[ti])
        if cnt == 0:
            continue
        s = R.gen_binary_key(2, n_prime)
        bsk = R.bsk_gen(s, S, t.k, t.N, t.l, t.beta, 0.0, 4)
        bskf = R.bsk_to_fourier(bsk)
        small = np.random.default_rng(0).integers(0, 2 ** 64, (threads, n_prime + 1), dtype=np.uint64)
        table = (np.arange(16, dtype=np.int64)) << 58
        R.pbs(small[:1], bskf, None, t.k, t.N, t.l, t.beta, table, 4, None, D)           # warm the FFT plan
        t0 = time.time()
        R.pbs(small, bskf, None, t.k, t.N, t.l, t.beta, table, 4, None, D)
        t_pbs = (time.time() - t0) * (t.n / n_prime)                                      # seconds for `threads` bootstraps
        ksk = np.random.default_rng(1).integers(0, 2 ** 64, (D, t.lk, n_prime + 1), dtype=np.uint64)
        big = np.random.default_rng(2).integers(0, 2 ** 64, (threads, D + 1), dtype=np.uint64)
        t0 = time.time()
        R.keyswitch(big, ksk, t.betak)
        t_ks = (time.time() - t0) * ((t.n + 1) / (n_prime + 1))
        per_ct = (t_pbs + t_ks) / threads
        detail[t.name] = dict(ms_per_pbs_per_core=per_ct * threads * 1e3 / 1.0, count=cnt)
        total_s += cnt * per_ct
    # ciphertext convolution: one 8-channel 3x3 layer on a 6x6 map, scaled by MACs
    cin, cout, hw = 8, 8, 6
    x = np.random.default_rng(3).integers(0, 2 ** 64, (cin, hw, hw, D + 1), dtype=np.uint64)
    w = np.random.default_rng(4).integers(-7, 8, (cout, cin, 3, 3)).astype(np.int32)
    t0 = time.time()
    R.conv2d(x, cin, hw, hw, D, w, 1, 1)
    macs = cout * cin * 9 * hw * hw
    total_s += (time.time() - t0) * (stats.conv_macs / macs)
    return dict(value=1.0 / total_s, unit="images/s", cores=threads, kind="port",
                sample=f"C twin (oracle/tfhe_ref.c, OpenMP {threads} threads): per tier {threads} bootstraps x {n_prime} of n blind-rotate "
                       f"iterations + key switch onto {n_prime + 1} of n+1 columns, scaled linearly in n; one 8x8x3x3 ciphertext conv scaled by MACs",
                s_per_image=total_s, per_tier=detail)


def _heartbeat(period=60.0):
    """one stderr line a minute: a long encrypted run (config #5 takes ~12 min) must not look hung to the job runner"""
    import threading
    t0 = time.time()

    def beat():
        while True:
            time.sleep(period)
            print(f"[bench] running, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()


def main():
    _heartbeat()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--batch-per-gpu", type=int, default=int(os.environ.get("DCTFHE_BENCH_BATCH", "4")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tier-policy", default="exact", choices=["exact", "p_error"],
                    help="exact (the metric): outputs equal the integer circuit; p_error: the reference-style stochastic regime, p_error=0.01 per look-up (speed only)")
    ap.add_argument("--rounding-method", default="exact", choices=["exact", "approximate"])
    ap.add_argument("--config", default="r20_24_16", choices=sorted(CONFIGS), help="BASELINE.json config; the metric is quoted on r20_24_16")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from dctfhe import models
    from dctfhe.quantized_module import compile_brevitas_qat_model
    from dctfhe.synthetic import synthetic_dct_batch

    B = args.batch_per_gpu
    # same circuit and same keys on every rank (seed-regenerated: no key traffic)
    factory, in_ch, img, make_batch, workload = CONFIGS[args.config]
    calib = make_batch(16 if args.config == "r18_48_112" else 100, 7)
    model = getattr(models, factory)(bit_width=4, in_channels=in_ch, img_size=img, seed=0)
    rtb = 6 if args.rounding_method == "exact" else {"n_bits": 6, "method": "approximate"}
    qm = compile_brevitas_qat_model(model, calib, n_bits=5, rounding_threshold_bits=rtb, p_error=0.01, device=local_rank, tier_policy=args.tier_policy)
    t0 = time.time()
    qm.fhe_circuit.keygen(seed=1)
    keygen_s = time.time() - t0
    stats = qm.statistics()

    # this rank's shard of the global synthetic batch: image i -> rank i % world
    from dctfhe.sharding import gather_in_image_order, shard_indices
    x_all = make_batch(B * world, 42)
    x = x_all[shard_indices(B * world, rank, world)]
    q = qm.quantize_input(x)
    phases = qm.encode_input(q)
    sess = qm._session("execute", B)
    cts = qm._keys.encrypt(phases.reshape(-1), 1000 + rank)
    t_up = time.time()
    sess.upload(cts)                      # inputs resident in HBM before the timed region
    upload_s = time.time() - t_up         # host -> device copy of the encrypted batch (reported, never part of `value`)
    input_bytes = cts.nbytes
    del cts

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        qm._ctx.synchronize()

    for _ in range(args.warmup):
        sess.run()
    sync()
    t0 = time.time()
    timings = []
    for _ in range(args.steps):
        timings.append(sess.run(timing=True))
    sync()
    elapsed = time.time() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # decrypt this shard, classify in the clear (reference utils.py:22), gather logits over RCCL
    out = sess.download().reshape(-1, qm._keys.D + 1)
    feats_q = qm.decode_output(qm._keys.decrypt(out).reshape(B, -1))
    clear_q = qm.forward_quantized(q, "disable")
    exact = bool(np.array_equal(feats_q, clear_q))
    diff = np.abs(feats_q.astype(np.int64) - clear_q.astype(np.int64))
    feats = torch.from_numpy(qm.dequantize_output(feats_q)).float()
    logits = feats @ torch.from_numpy(model.classifier_w).float().T + torch.from_numpy(model.classifier_b).float()
    if world > 1:
        all_logits = gather_in_image_order(logits.cuda(), world).cpu()      # one RCCL all_gather, global image order
        flags = torch.tensor([1.0 if exact else 0.0], device="cuda")
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        exact = bool(flags.item() > 0.5)
    else:
        all_logits = logits

    if rank == 0:
        images = B * world * args.steps
        value = images / elapsed
        ps = qm.compiled.param_set
        # dominant kernel: the bootstrap of the tier with the most time
        tm = timings[-1]
        pbs_ms = [sum(t.pbs_ms[i] for t in timings) for i in range(len(ps.tiers))]
        dom = int(np.argmax(pbs_ms))
        td = ps.tiers[dom]
        launches = sum(t.pbs_launches[dom] for t in timings)
        cts_dom = stats.pbs_count[dom] * B * args.steps
        avg_launch_s = pbs_ms[dom] * 1e-3 / max(launches, 1)
        cts_per_launch = cts_dom / max(launches, 1)
        N = td.N
        unroll = getattr(td, "unroll", 1)
        bsk_bytes = (3 * td.n // 2 if unroll == 2 else td.n) * td.l * (td.k + 1) ** 2 * N * 8.0
        alg_bytes = cts_per_launch * ((td.n + 1) * 8.0 + (ps.D + 1) * 8.0) + bsk_bytes
        M = N / 2
        fft = 5.0 * M * math.log2(M)
        if unroll == 2:   # two-bit blind rotation: per pair the same transforms, 3 key blocks folded with their monomials
            flops_per_pbs = (td.n / 2) * ((td.k + 1) * td.l * fft + (td.k + 1) * fft + (td.k + 1) ** 2 * td.l * M * 30.0 + M * 18.0)
        else:
            flops_per_pbs = td.n * ((td.k + 1) * td.l * fft + (td.k + 1) * fft + (td.k + 1) ** 2 * td.l * M * 8.0)
        fp64_live = qm._ctx.fp64_peak()
        achieved_gbs = alg_bytes / avg_launch_s / 1e9
        achieved_tf = flops_per_pbs * cts_per_launch / avg_launch_s / 1e12
        res = {
            "metric": "encrypted images/sec, ResNet-20 DCT-24x16^2 CIFAR-10" if args.config == "r20_24_16" else f"encrypted images/sec, {args.config}",
            "value": value,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 torus + f64 FFT",
            "data": "synthetic",
            "config": {"workload": f"{workload}, {B} encrypted image(s) per GPU, " +
                                   ("exact-evaluation tiers" if args.tier_policy == "exact" else "p_error=0.01 tiers (stochastic outputs, speed only)") +
                                   (", approximate rounding" if args.rounding_method == "approximate" else "") + ", rounding_threshold_bits=6, n_bits=5, bit_width=4",
                       "images_per_gpu": B, "global_batch": B * world, "parallelism": f"image-sharded x{world}",
                       "s_per_image_per_gpu": elapsed / (B * args.steps),
                       "pbs_per_image": int(sum(stats.pbs_count)), "bit_steps_per_image": int(stats.bit_steps),
                       "table_lookups_per_image": int(stats.lut_sites), "conv_macs_per_image": int(stats.conv_macs),
                       "max_bit_width": int(stats.max_bit_width), "keygen_s": keygen_s,
                       "input_upload_s": upload_s, "input_bytes_per_gpu": int(input_bytes),
                       "images_per_s_pcie_inclusive": images / (elapsed + upload_s * args.steps),
                       "bit_exact_vs_integer_circuit": exact, "tier_policy": args.tier_policy, "rounding_method": args.rounding_method,
                       "outputs_equal_frac": float((diff == 0).mean()), "outputs_max_abs_diff": int(diff.max()),
                       "expected_boundary_flips_per_image": float(getattr(qm.compiled, "expected_boundary_flips_per_image", 0.0)),
                       "predicted_labels": all_logits.argmax(dim=1).tolist(),
                       "expected_table_failures_per_image": qm.compiled.expected_failures_per_image},
            "roofline": {"bound": "hbm", "kernel": f"pbs_kernel<logN={td.logN},k={td.k},l={td.l}> (tier {td.name})",
                         "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": measured_hbm_traffic(f"pbs_kernel<{td.logN},{td.k},{td.l},", cts_per_launch), "avg_launch_ms": avg_launch_s * 1e3, "cts_per_launch": cts_per_launch,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "the blind rotate is f64-VALU/LDS bound (SURVEY 8d); see roofline_fp64 for the bounding roof"},
            "roofline_fp64": {"bound": "fp64_valu", "achieved": achieved_tf, "peak": FP64_SPEC_TFLOPS, "unit": "TFLOP/s",
                              "frac": achieved_tf / FP64_SPEC_TFLOPS, "peak_live_fma_probe": fp64_live,
                              "frac_of_live_probe": achieved_tf / fp64_live, "flops_per_bootstrap": flops_per_pbs},
            "time_split_ms": {"total": sum(t.total_ms for t in timings), "linear": sum(t.linear_ms for t in timings),
                              "keyswitch": sum(t.ks_ms for t in timings),
                              "pbs_by_tier": {ps.tiers[i].name: pbs_ms[i] for i in range(len(ps.tiers))}},
            "algorithmic": {"bytes_per_image": stats.bytes_algorithmic, "key_bytes_per_pass": stats.key_bytes_per_pass,
                            "flops_f64_per_image": stats.flops_f64,
                            "hbm_frac_whole_pipeline": (stats.bytes_algorithmic + stats.key_bytes_per_pass / B) * B * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                            "fp64_frac_whole_pipeline": stats.flops_f64 * B * args.steps / elapsed / 1e12 / FP64_SPEC_TFLOPS},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(qm, stats)
                res["cpu_baseline"]["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
            except Exception as e:  # the baseline is a report, never the product path
                res["cpu_baseline"] = {"error": repr(e)}
        res["reference_published_s_per_image"] = 565.0      # README.md:84, 96-core CPU; an anchor, not a vs_baseline (other hardware)
        print(json.dumps(res))
    qm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
