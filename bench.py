#!/usr/bin/env python
"""bench.py -- encrypted images/s of the DCT-CryptoNets hot path on N MI355X GPUs.

Metric (BASELINE.json): encrypted images/sec (+ s/image), ResNet-20 DCT 24x16^2 CIFAR-10 trunk.
A "step" is one pass of the homomorphic circuit (dctfhe_session_run: every conv, add, exact rounding,
key switch and bootstrap of the trunk) over one batch of synthetic encrypted images that is already
resident in HBM (reference timed loop: homomorphic_eval.py:350-361, `elapsed / test_subset`).  The batch
shards by image across ranks (one process per GPU); there is no collective on the data path -- the only
exchange is one RCCL all_gather of the decrypted-side logits at the end.

Usage:  python bench.py --gpus N --steps K --warmup W [--batch-per-gpu B] [--budget-s S]
  * N > 1 without WORLD_SIZE in the environment: bench.py starts its own N ranks (child processes of
    `python -m torch.distributed.run`, before anything here touches a GPU) and exits with their code;
    under torch.distributed.run it is one of the ranks.
  * One encrypted image takes seconds, so K and W are CAPPED by a wall-clock budget (--budget-s, default 450 s from
    process start, env DCTFHE_BENCH_BUDGET_S): after the first pass the loop keeps as many of the requested passes
    as fit; the JSON reports the steps / warm-up passes actually run (and `requested`).  When not even a second pass
    fits (multi-image configs whose pass takes minutes), the first pass -- bracketed like a step -- is the timed step.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import signal
import subprocess
import sys
import threading
import time

T_START = time.time()
ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dct-cryptonets_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_SPEC_TFLOPS = 78.6          # MI355X vector FP64 (spec; the guide does not list it, so the live FMA probe is reported beside it)


# VALU instructions per blind-rotate loop iteration and wave, and waves per ciphertext, of the bootstrap kernels (ISA listing of the
# shipped build: hipcc -S + tools/isa_hist.py).  key: (logN, k, l, unroll).  An iteration consumes `unroll` key bits.
VALU_PER_ITERATION = {(13, 1, 1, 2): (2275, 8), (12, 1, 1, 2): (2203, 4), (11, 1, 1, 2): (2156, 2), (11, 1, 3, 1): (4329, 2), (11, 1, 3, 2): (5968, 2), (10, 2, 1, 1): (2579, 1),
                      (10, 2, 1, 2): (3589, 1), (10, 2, 2, 1): (4520, 1)}      # final round-3 build (tools/isa_hist.py: f64 + other VALU of the loop body)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4          # wave-instructions per second: 256 CUs x 4 SIMDs, one f64 wave instruction per 4 cycles, 2.4 GHz
N_CU, SPEC_CLOCK_HZ = 256, 2.4e9
L1_DELIVERY_BYTES_PER_CLK_CU = 64.0            # a CU's vector L1 returns one 64-byte half line per clock to the registers (a wave's 16 B/lane load: 16 clocks);
                                               # measured here: 60 B/clk in the key phases of the three-level kernel (profiles/r03_exp_phase_clock.log)


def _dct_batch(n, seed):
    from dctfhe.synthetic import synthetic_dct_batch
    return synthetic_dct_batch(n, seed=seed)


def _rgb_batch(n, seed):
    import numpy as np
    from dctfhe import frontend, synthetic
    tf = frontend.rgb_eval_transform(32)
    return np.stack([tf(im) for im in synthetic.synthetic_images(n, seed)]).astype(np.float32)


def _dct8_112_batch(n, seed):
    """config #5: 8x8 JPEG-domain DCT, 48 channels, 112x112: Resize(1030) -> CenterCrop(896) -> 8x8 DCT"""
    import numpy as np
    from dctfhe import frontend, synthetic
    tf = frontend.dct_eval_transform(filter_size=8, image_size_dct=112, channels=48)
    return np.stack([tf(im) for im in synthetic.synthetic_images(n, seed, size=96)]).astype(np.float32)


# name -> (model factory name, in_channels, img_size, input batch maker, description)
CONFIGS = {
    "r20_24_16": ("ResNet20QAT", 24, 16, _dct_batch, "ResNet-20 24x16^2 DCT CIFAR-10 trunk (BASELINE config #1/#2 shape)"),
    "r20_3_32": ("ResNet20QAT", 3, 32, _rgb_batch, "ResNet-20 3x32^2 RGB CIFAR-10 trunk (BASELINE config #3)"),
    "r18_3_32": ("ResNet18QAT", 3, 32, _rgb_batch, "ResNet-18 3x32^2 RGB CIFAR-10 trunk (BASELINE config #4 shape)"),
    "r18_48_112": ("ResNet18QAT", 48, 112, _dct8_112_batch, "ResNet-18 48x112^2 DCT ImageNet trunk (BASELINE config #5 shape; 16 calibration images)"),
}


def measured_hbm_traffic(kernel_tag, cts_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (tools/rocprof_db_summary.py; FETCH_SIZE and
    WRITE_SIZE need their own rocprofv3 passes, so they cannot be collected inside this run).  Scaled by ciphertexts per
    launch; None when the summary has no entry for this kernel."""
    for name in ("r03_pmc_hbm.json", "r02_pmc_hbm.json", "r01_pmc_hbm.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        prof = json.load(open(path))
        for k, v in prof.items():
            if kernel_tag in k.replace(" ", "") and v.get("launches"):
                per_ct = v["hbm_bytes_per_launch"] / v.get("cts_per_launch", cts_per_launch)
                return per_ct * cts_per_launch
    return None


def conv_macs_of(compiled):
    macs = 0
    for o in compiled.ops:
        if o.type == 1:
            s, d = compiled.tensors[o.src0], compiled.tensors[o.dst]
            macs += d.C * d.H * d.W * s.C * o.ip[1] * o.ip[2]
    return macs


def cpu_baseline(compiled, max_threads=16, target_s_per_tier=4.0):
    """Times the CPU oracle (oracle/tfhe_ref.c, the C twin; kind "port" -- the reference's own CPU path lives in absent
    third-party wheels) on a bounded sample of the same workload and scales by the circuit's counts.  Sample, per parameter
    tier the circuit uses: `threads x reps` FULL bootstraps (all n blind-rotate iterations; the key is random numbers of the
    right shape, timing does not depend on its values) and as many key switches over the effective input dimension; plus one
    3x3 ciphertext convolution, scaled by MACs.  About 15-25 s of CPU work."""
    import numpy as np
    from oracle import ref_loader as R
    R.build()
    R.lib().ref_set_num_threads(min(max_threads, os.cpu_count() or 1))
    ps = compiled.param_set
    threads = R.lib().ref_num_threads()
    D = ps.D
    deff = ps.input_dim or D
    counts = compiled.pbs_counts()
    total_s, detail = 0.0, {}
    rng = np.random.default_rng(0)
    t_begin = time.time()
    for t in ps.tiers:
        cnt = int(counts.get(t.name, 0))
        if cnt == 0:
            continue
        rows = (t.k + 1) * t.l
        bskf = rng.standard_normal((t.n, rows, t.k + 1, t.N))
        table = (np.arange(16, dtype=np.int64)) << 58
        small = rng.integers(0, 2 ** 64, (threads, t.n + 1), dtype=np.uint64)
        t0 = time.time()
        R.pbs(small, bskf, None, t.k, t.N, t.l, t.beta, table, 4, None, D)               # also warms the FFT plan
        one = time.time() - t0
        reps = int(max(1, min(32, target_s_per_tier / max(one, 1e-3))))
        small = rng.integers(0, 2 ** 64, (threads * reps, t.n + 1), dtype=np.uint64)
        t0 = time.time()
        R.pbs(small, bskf, None, t.k, t.N, t.l, t.beta, table, 4, None, D)
        t_pbs = time.time() - t0
        del bskf
        ksk = rng.integers(0, 2 ** 64, (deff, t.lk, t.n + 1), dtype=np.uint64)
        big = rng.integers(0, 2 ** 64, (threads * reps, deff + 1), dtype=np.uint64)
        t0 = time.time()
        R.keyswitch(big, ksk, t.betak)
        t_ks = time.time() - t0
        del ksk
        per_ct = (t_pbs + t_ks) / (threads * reps)              # wall seconds per ciphertext with all threads busy
        detail[t.name] = dict(ms_per_pbs_one_core=t_pbs / reps * 1e3, ms_per_keyswitch_one_core=t_ks / reps * 1e3, count=cnt, sample_cts=threads * reps)
        total_s += cnt * per_ct
    cin, cout, hw = 8, 8, 6
    x = rng.integers(0, 2 ** 64, (cin, hw, hw, deff + 1), dtype=np.uint64)
    w = rng.integers(-7, 8, (cout, cin, 3, 3)).astype(np.int32)
    t0 = time.time()
    R.conv2d(x, cin, hw, hw, deff, w, 1, 1)
    macs = cout * cin * 9 * hw * hw
    total_s += (time.time() - t0) * (conv_macs_of(compiled) / macs)
    return dict(value=1.0 / total_s, unit="images/s", cores=threads, kind="port",
                sample=f"C twin (oracle/tfhe_ref.c, OpenMP {threads} threads): per tier threads x reps full bootstraps (all n blind-rotate "
                       f"iterations) + key switches over {deff} mask words, scaled by the circuit's counts; one 8x8x3x3 ciphertext conv scaled by MACs",
                s_per_image=total_s, sample_wall_s=time.time() - t_begin, per_tier=detail)


def _heartbeat(state, period=30.0):
    """one stderr line every 30 s: a long encrypted run must not look hung to the job runner, and a killed run leaves its progress"""
    def beat():
        while True:
            time.sleep(period)
            print(f"[bench] {time.time() - T_START:.0f} s: {state.get('phase', '?')}, passes done {state.get('passes', 0)}"
                  + (f", last pass {state['last_s']:.2f} s" if "last_s" in state else ""), file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()


def plan_passes(req_warm, req_steps, warm_done, steps_done, pass_s, remaining_s):
    """How many more warm-up passes and timed steps fit into `remaining_s` at `pass_s` per pass.  Timed steps take priority over
    warm-up; at least one timed step always runs.  -> (warm_total, steps_total)."""
    fit = 10 ** 9 if math.isinf(remaining_s) else int(max(0.0, remaining_s) // max(pass_s, 1e-6))      # --budget-s 0: no cap
    want_steps = max(0, req_steps - steps_done)
    want_warm = max(0, req_warm - warm_done) if steps_done == 0 else 0
    more_steps = min(want_steps, fit)
    more_warm = min(want_warm, max(0, fit - more_steps))
    if steps_done + more_steps == 0:
        more_steps = 1
    return warm_done + more_warm, steps_done + more_steps


def first_pass_is_the_step(first_s, remaining_s, req_warm, req_steps, interrupted=False):
    """The first pass is bracketed and timed like a step.  It IS the timed step when nothing else was asked for (--warmup 0 --steps 1),
    when the run was interrupted, or when what is left of the budget does not hold another pass with 15 % to spare -- a multi-image
    config whose one pass takes minutes then ends inside the driver's wall instead of running a warm-up and a timed pass back to back."""
    return bool(interrupted or (req_warm == 0 and req_steps <= 1) or remaining_s < 1.15 * first_s)


def self_launch(args_list, n, module="torch.distributed.run"):
    """`python bench.py --gpus N` outside torch.distributed.run: start N ranks as CHILD processes (never an exec of a
    process that has touched the GPU -- nothing here has) and hand back their exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", module, "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(args_list)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["DCTFHE_BENCH_T0"] = repr(T_START)      # the children budget against the launcher's clock
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=int(os.environ.get("DCTFHE_BENCH_BATCH", "1")),
                    help="encrypted images per GPU and step (default 1 = the reference's execute-mode case, run_homomorphic_eval.sh:22-23)")
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("DCTFHE_BENCH_BUDGET_S", "450")),
                    help="wall-clock budget from process start; steps/warm-up are capped to fit (0 = no cap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tier-policy", default="exact", choices=["exact", "p_error"],
                    help="exact (the metric): outputs equal the integer circuit; p_error: the reference-style stochastic regime, p_error=0.01 per look-up (speed only)")
    ap.add_argument("--rounding-method", default="exact", choices=["exact", "approximate"])
    ap.add_argument("--config", default="r20_24_16", choices=sorted(CONFIGS), help="BASELINE.json config; the metric is quoted on r20_24_16")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 rehearsal on a single-GPU box: every rank evaluates on GPU 0, collectives over gloo (same code path "
                         "otherwise; the numbers mean nothing)")
    ap.add_argument("--launch-check", action="store_true",
                    help="launcher self-test (CPU, gloo): ranks shard a fake batch, all_gather it and rank 0 prints a JSON line; no GPU work")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    t_start = float(os.environ.get("DCTFHE_BENCH_T0", T_START))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import torch.distributed as dist
    from dctfhe.sharding import agree_min, all_true, barrier, broadcast_seed, gather_in_image_order, max_over_ranks, shard_indices

    if args.launch_check:
        dist.init_process_group("gloo")
        B = args.batch_per_gpu
        fake = torch.arange(B * world * 4, dtype=torch.float32).reshape(B * world, 4)
        got = gather_in_image_order(fake[shard_indices(B * world, rank, world)], world)
        ok = bool(torch.equal(got, fake))
        flag = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "gather_in_image_order": bool(flag.item() > 0.5)}), flush=True)
        dist.destroy_process_group()
        return

    state = {"phase": "setup", "passes": 0}
    _heartbeat(state)
    gpu = 0 if args.rehearse_on_one_gpu else local_rank
    cdev = torch.device("cpu") if args.rehearse_on_one_gpu else torch.device("cuda", gpu)       # where the collectives' tensors live
    torch.cuda.set_device(gpu)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=cdev)

    from dctfhe import models
    from dctfhe.quantized_module import compile_brevitas_qat_model

    B = args.batch_per_gpu
    # same circuit and same keys on every rank (seed-regenerated: no key traffic)
    factory, in_ch, img, make_batch, workload = CONFIGS[args.config]
    calib = make_batch(16 if args.config == "r18_48_112" else 100, 7)
    model = getattr(models, factory)(bit_width=4, in_channels=in_ch, img_size=img, seed=0)
    rtb = 6 if args.rounding_method == "exact" else {"n_bits": 6, "method": "approximate"}
    t0 = time.time()
    qm = compile_brevitas_qat_model(model, calib, n_bits=5, rounding_threshold_bits=rtb, p_error=0.01, device=gpu, tier_policy=args.tier_policy)
    compile_s = time.time() - t0

    # CPU baseline: rank 0, N = 1 only, on its own thread while the GPU warms up (ctypes drops the GIL); joined before the timed region
    cpu_box = {}
    cpu_thread = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        def _cpu():
            try:
                cpu_box["res"] = cpu_baseline(qm.compiled)
            except Exception as e:  # the baseline is a report, never the product path
                cpu_box["res"] = {"error": repr(e)}
        cpu_thread = threading.Thread(target=_cpu, daemon=True)
        cpu_thread.start()

    state["phase"] = "keygen"
    t0 = time.time()
    # one 256-bit key seed for the whole job: rank 0 draws it from the OS and broadcasts it; every rank regenerates the same
    # client + evaluation keys from it on its own GPU (no key traffic) and encrypts from its own counter range
    qm.fhe_circuit.keygen(seed=broadcast_seed(os.urandom(32), world, cdev))
    # (every rank's client handle draws its own 128-bit encryption nonce from the OS: ranks sharing the seed never share masks or noise)
    keygen_s = time.time() - t0
    stats = qm.statistics()
    # seeded classifier with logits centred on the calibration features (clear circuit): labels differ between images
    from dctfhe.synthetic import centre_classifier
    centre_classifier(model, qm.forward(calib[:32], fhe="disable"))

    # this rank's shard of the global synthetic batch: image i -> rank i % world.  Client-side stages are timed one by one for the
    # reference-style end-to-end figure (homomorphic_eval.py:350-361 times transform + quantise + encrypt + run + decrypt + classifier)
    t_fe = time.time()
    x_all = make_batch(B * world, 42)
    frontend_s = (time.time() - t_fe) / world          # front-end of this rank's share (the batch maker transforms every image)
    x = x_all[shard_indices(B * world, rank, world)]
    t_q = time.time()
    q = qm.quantize_input(x)
    phases = qm.encode_input(q)
    quantize_s = time.time() - t_q
    sess = qm._session("execute", B)
    in_dim, out_dim = sess.dims()         # compact wire form: rows of input_dim + 1 / last ring + 1 words, not D + 1
    t_enc = time.time()
    cts = qm._keys.encrypt(phases.reshape(-1), in_dim)
    encrypt_s = time.time() - t_enc
    t_up = time.time()
    sess.upload(cts, in_dim)              # inputs resident in HBM before the timed region
    upload_s = time.time() - t_up         # host -> device copy of the encrypted batch (reported, never part of `value`)
    input_bytes = cts.nbytes
    del cts

    def sync():
        barrier(world)
        torch.cuda.synchronize()
        qm._ctx.synchronize()

    def agree(warm, steps):
        """every rank runs the same number of passes: the minimum over ranks"""
        w, st = agree_min([warm, steps], world, cdev)
        return w, st

    # ---- wall-clock budget -------------------------------------------------------------------------------------------------
    reserve_s = 25.0                     # decrypt + clear circuit + JSON + teardown
    budget = args.budget_s if args.budget_s > 0 else float("inf")

    def remaining():
        return budget - (time.time() - t_start) - reserve_s

    interrupted = {"flag": False}

    def on_term(signum, frame):          # a runner's SIGTERM: report what has been measured instead of dying silently
        interrupted["flag"] = True
    signal.signal(signal.SIGTERM, on_term)

    # ---- passes.  The FIRST pass is bracketed and timed like a step (barrier + synchronize on both sides).  If what is left of the
    # budget after it does not hold another pass, it IS the timed step (reported: first_pass_counted) -- a multi-image config whose one
    # pass takes minutes (ResNet-18 3x32^2 with 8 images per GPU: ~5 min) then still ends inside the driver's wall instead of running a
    # warm-up pass and a timed pass back to back.  Otherwise it is the first warm-up pass and the plan for the rest is made from it.
    state["phase"] = "first pass"
    sync()
    t_first = time.time()
    first_timing = sess.run(timing=True)
    sync()
    first_s = max_over_ranks(time.time() - t_first, world, cdev)
    state.update(passes=1, last_s=first_s)
    timings = []
    warm_done, steps_done = 0, 0
    # every rank takes the same branch: the first pass counts as soon as ONE rank says so
    first_pass_counted = not all_true(not first_pass_is_the_step(first_s, remaining(), args.warmup, args.steps, interrupted["flag"]), world, cdev)
    if first_pass_counted:
        timings.append(first_timing)
        steps_done, elapsed = 1, first_s
    else:
        if args.warmup == 0:
            # no warm-up asked for: the first pass is step 1 of the timed region; the remaining steps follow and are timed together
            timings.append(first_timing)
            steps_done = 1
            _, steps_total = agree(*plan_passes(0, args.steps, 0, 1, first_s, remaining()))
            warm_total = 0
        else:
            warm_done = 1
            warm_total, steps_total = agree(*plan_passes(args.warmup, args.steps, 1, 0, first_s, remaining()))
        state["phase"] = "warm-up"
        while warm_done < warm_total and not interrupted["flag"]:
            t0 = time.time()
            sess.run()
            warm_done += 1
            state.update(passes=warm_done, last_s=time.time() - t0)
    if cpu_thread is not None:
        state["phase"] = "waiting for the CPU baseline"
        cpu_thread.join()
    if not first_pass_counted:
        state["phase"] = "timed steps"
        sync()
        t0 = time.time()
        while steps_done < steps_total and not interrupted["flag"]:
            ts = time.time()
            timings.append(sess.run(timing=True))
            steps_done += 1
            state.update(passes=warm_done + steps_done, last_s=time.time() - ts)
        sync()
        elapsed = max_over_ranks(time.time() - t0, world, cdev) + (first_s if args.warmup == 0 else 0.0)
    state["phase"] = "decrypt + check"

    # decrypt this shard, classify in the clear (reference utils.py:22), gather logits over RCCL
    t_dl = time.time()
    out = sess.download(out_dim).reshape(-1, out_dim + 1)
    download_s = time.time() - t_dl
    output_bytes = out.nbytes
    t_dec = time.time()
    feats_q = qm.decode_output(qm._keys.decrypt(out, out_dim).reshape(B, -1))
    decrypt_s = time.time() - t_dec
    t_cls = time.time()
    feats = torch.from_numpy(qm.dequantize_output(feats_q)).float()
    logits = feats @ torch.from_numpy(model.classifier_w).float().T + torch.from_numpy(model.classifier_b).float()
    classifier_s = time.time() - t_cls
    clear_q = qm.forward_quantized(q, "disable")
    exact = bool(np.array_equal(feats_q, clear_q))
    diff = np.abs(feats_q.astype(np.int64) - clear_q.astype(np.int64))
    if world > 1:
        all_logits = gather_in_image_order(logits.to(cdev), world).cpu()      # one RCCL all_gather, global image order
        exact = all_true(exact, world, cdev)
    else:
        all_logits = logits

    if rank == 0:
        steps = max(steps_done, 1)
        images = B * world * steps
        value = images / elapsed
        ps = qm.compiled.param_set
        # dominant kernel: the bootstrap of the tier with the most time
        pbs_ms = [sum(t.pbs_ms[i] for t in timings) for i in range(len(ps.tiers))]
        dom = int(np.argmax(pbs_ms))
        td = ps.tiers[dom]
        launches = sum(t.pbs_launches[dom] for t in timings)
        cts_dom = stats.pbs_count[dom] * B * steps
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
        step_s = [t.total_ms * 1e-3 for t in timings]
        wall_now = time.time() - t_start
        res = {
            "metric": "encrypted images/sec, ResNet-20 DCT-24x16^2 CIFAR-10" if args.config == "r20_24_16" else f"encrypted images/sec, {args.config}",
            "value": value,
            "unit": "images/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warm_done,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 torus + f64 FFT (one-level tiers keep 32-bit accumulators)",
            "data": "synthetic",
            "requested": {"steps": args.steps, "warmup": args.warmup, "budget_s": args.budget_s,
                          "capped_by_budget": bool(steps < args.steps or warm_done < args.warmup), "interrupted": interrupted["flag"],
                          "first_pass_counted": bool(first_pass_counted), "first_pass_s": first_s},
            "s_per_image": elapsed / (B * steps),
            "config": {"workload": f"{workload}, {B} encrypted image(s) per GPU and step, " +
                                   ("exact-evaluation tiers" if args.tier_policy == "exact" else "p_error=0.01 tiers (stochastic outputs, speed only)") +
                                   (", approximate rounding" if args.rounding_method == "approximate" else "") + ", rounding_threshold_bits=6, n_bits=5, bit_width=4",
                       "images_per_gpu": B, "global_batch": B * world, "parallelism": f"image-sharded x{world}",
                       "s_per_image_per_gpu": elapsed / (B * steps),
                       "step_s_min_max": [min(step_s), max(step_s)] if step_s else None,
                       "pbs_per_image": int(sum(stats.pbs_count)), "bit_steps_per_image": int(stats.bit_steps),
                       "table_lookups_per_image": int(stats.lut_sites), "conv_macs_per_image": int(stats.conv_macs),
                       "max_bit_width": int(stats.max_bit_width), "compile_s": compile_s, "keygen_s": keygen_s,
                       "key_seed": "32 bytes of os.urandom on rank 0, broadcast; each rank's handle draws its own 128-bit encryption nonce",
                       "input_upload_s": upload_s, "input_bytes_per_gpu": int(input_bytes), "output_bytes_per_gpu": int(output_bytes),
                       "wire_format": f"compact rows: {in_dim} + 1 words per input ciphertext, {out_dim} + 1 per output (D = {ps.D})",
                       "images_per_s_pcie_inclusive": images / (elapsed + upload_s * steps),
                       "bit_exact_vs_integer_circuit": exact, "tier_policy": args.tier_policy, "rounding_method": args.rounding_method,
                       "outputs_equal_frac": float((diff == 0).mean()), "outputs_max_abs_diff": int(diff.max()),
                       "expected_boundary_flips_per_image": float(getattr(qm.compiled, "expected_boundary_flips_per_image", 0.0)),
                       "predicted_labels": all_logits.argmax(dim=1).tolist(),
                       "expected_table_failures_per_image": qm.compiled.expected_failures_per_image},
            "roofline": {"bound": "hbm", "binding": "roofline_fp64", "binding_frac": achieved_tf / FP64_SPEC_TFLOPS,
                         "kernel": f"pbs_kernel<logN={td.logN},k={td.k},l={td.l}> (tier {td.name})",
                         "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": measured_hbm_traffic(f"pbs_kernel<{td.logN},{td.k},{td.l},", cts_per_launch), "avg_launch_ms": avg_launch_s * 1e3, "cts_per_launch": cts_per_launch,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "HBM is NOT the roof this kernel sits under: the blind rotate is f64-VALU / LDS bound (SURVEY 8d). The binding roof is "
                                 "roofline_fp64 (same launch, same measured duration); `binding_frac` repeats its fraction here"},
            "roofline_fp64": {"bound": "fp64_valu", "achieved": achieved_tf, "peak": FP64_SPEC_TFLOPS, "unit": "TFLOP/s",
                              "frac": achieved_tf / FP64_SPEC_TFLOPS, "peak_live_fma_probe": fp64_live,
                              "frac_of_live_probe": achieved_tf / fp64_live, "flops_per_bootstrap": flops_per_pbs},
            "roofline_valu_issue": None,
            "time_split_ms": {"total": sum(t.total_ms for t in timings), "linear": sum(t.linear_ms for t in timings),
                              "keyswitch": sum(t.ks_ms for t in timings),
                              "pbs_by_tier": {ps.tiers[i].name: pbs_ms[i] for i in range(len(ps.tiers))}},
            "algorithmic": {"bytes_per_image": stats.bytes_algorithmic, "key_bytes_per_pass": stats.key_bytes_per_pass,
                            "flops_f64_per_image": stats.flops_f64,
                            "hbm_frac_whole_pipeline": (stats.bytes_algorithmic + stats.key_bytes_per_pass / B) * B * steps / elapsed / 1e9 / HBM_PEAK_GBS,
                            "fp64_frac_whole_pipeline": stats.flops_f64 * B * steps / elapsed / 1e12 / FP64_SPEC_TFLOPS},
            # reference-style end to end (homomorphic_eval.py:350-361: DataLoader transform + forward(quantise, encrypt, run, decrypt,
            # dequantise) + clear classifier, `elapsed / test_subset`), this rank's shard, every stage measured in this run
            "end_to_end": {"frontend_s": frontend_s, "quantize_s": quantize_s, "encrypt_s": encrypt_s, "upload_s": upload_s,
                           "run_s_per_step": elapsed / steps, "download_s": download_s, "decrypt_s": decrypt_s, "classifier_s": classifier_s,
                           "e2e_s_per_image": (frontend_s + quantize_s + encrypt_s + upload_s + elapsed / steps + download_s + decrypt_s + classifier_s) / B,
                           "note": "client stages on the host CPU (numpy front-end) and the GPU (encrypt / decrypt kernels), PCIe both ways included"},
            "consistency": {"wall_since_start_s": wall_now, "timed_s": elapsed, "fits_in_driver_run": bool(elapsed <= wall_now)},
        }
        vi = VALU_PER_ITERATION.get((td.logN, td.k, td.l, unroll))
        if vi:      # the roof these kernels actually sit under: f64 instruction issue (adds and multiplies fill as many slots as FMAs)
            instr = vi[0] * vi[1] * (td.n / unroll) * cts_per_launch
            res["roofline_valu_issue"] = {"bound": "valu_issue", "achieved": instr / avg_launch_s / 1e9, "peak": VALU_ISSUE_PEAK / 1e9, "unit": "G wave-instr/s",
                                          "frac": instr / avg_launch_s / VALU_ISSUE_PEAK, "valu_instr_per_iteration_per_wave": vi[0], "waves_per_ciphertext": vi[1],
                                          # what the power limit lets a pure f64 FMA stream issue on this box, in wave instructions per second
                                          "peak_live_fma_probe": fp64_live * 1e12 / 2 / 64 / 1e9, "frac_of_live_probe": instr / avg_launch_s / (fp64_live * 1e12 / 2 / 64),
                                          "note": "instruction counts from the ISA of the shipped build; peak at the 2.4 GHz spec clock (the kernels hold 2.04-2.38 GHz)"}
        # The other roof, found with the phase clock of tools/exp_pbs.hip (profiles/r03_exp_phase_clock.log): every ciphertext pulls the whole
        # Fourier key through its CU's vector L1 once -- bsk_bytes per ciphertext, never reused inside the CU -- and a CU's L1 hands at most
        # 64 bytes per clock to its registers (42-51 measured: profiles/r03_exp_stream_l1_delivery.log).  The key-product phases of a bootstrap
        # run at that ceiling and the transform phases at the f64 issue rate, one after the other for the waves of a workgroup that share
        # its barriers: the two fractions add up to most of the time.
        l1_peak = L1_DELIVERY_BYTES_PER_CLK_CU * N_CU * SPEC_CLOCK_HZ
        l1_bytes = bsk_bytes * cts_per_launch
        res["roofline_l1_delivery"] = {"bound": "vector_l1_to_registers", "achieved": l1_bytes / avg_launch_s / 1e12, "peak": l1_peak / 1e12, "unit": "TB/s",
                                       "frac": l1_bytes / avg_launch_s / l1_peak, "bytes_per_bootstrap": bsk_bytes,
                                       "note": "64 B/clk/CU x 256 CUs at the 2.4 GHz spec clock; key bytes per bootstrap x bootstraps per launch / the same measured launch duration. "
                                               "frac + roofline_valu_issue.frac ~ 0.9-1: the key phases (L1-bound) and the transform phases (issue-bound) of the "
                                               "lock-step waves of a CU do not overlap (DESIGN.md section 5)"}
        if cpu_box.get("res") is not None:
            res["cpu_baseline"] = cpu_box["res"]
            if "value" in res["cpu_baseline"]:
                res["cpu_baseline"]["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
        res["reference_published_s_per_image"] = 565.0      # README.md:84, 96-core CPU; an anchor, not a vs_baseline (other hardware)
        print(json.dumps(res), flush=True)
    qm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
