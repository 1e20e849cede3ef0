#!/usr/bin/env python
"""bench.py — sequences/sec of ONE EasyDGL optimizer step (forward + backward + TF-Adam) on the headline
configuration of BASELINE.json: per-GPU batch 512, seqslen 100 (T=101 positions), num_units 128, 8 heads,
1 block, num_items 20000 (I=20001 table rows), masklen 20, 16 mark types, dropout 0.1/0.1, bf16 activations
with f32 accumulation / master weights.  Synthetic data, random-init weights.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

(`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run with N ranks; a launcher whose
WORLD_SIZE differs from --gpus is an error.)

Rank 0 prints ONE JSON line.  The timed steps rotate through 8 distinct synthetic batches resident in HBM (the engine is pointed
at them: no copies).  `value` = N * 512 * K / (max-over-ranks wall time of the K timed steps); `step_ms_hipevents` = median /
p10 / p90 of HIP-event times of groups of 5 steps of the same region.  `roofline` is the dominant kernel's achieved MFMA rate —
strip_kernel<ROLE_YF>, the row-side pass of the fused scoring / cross-entropy: ALGORITHMIC FLOPs per launch (both of its products,
at the mean weighted-row count of the bracketed batches) over the mean launch time measured with HIP events on the launch stream
inside the timed region; `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/).
`roofline_attention` does the same for the K3 BiMAU forward launch and its three backward passes (the brackets rotate: one
kernel id per step), with the PMC pipe utilisation of those kernels beside it.  `cpu_baseline` times the restated reference graph
(oracle/torch_ref.py, float32, reference op order) on this host's cores at the same batch of 512 on a bounded sample;
`cpu_baseline_1thread_batch16` is the reference's own single-thread setting.  At N = 1 `extras` adds the secondary rows of SURVEY
§8(d): masklen 6, all rows weighted, dropout off, multi-hot marks, uniform item ids, the device masker inside the step, the
published recipe's shape, the config-3 K1 encode line with Zipf and uniform ids (algorithmic and counter-side GB/s) and the
sharded-eval step.

    python bench.py --workload recipe            # the published recipe's shape (runme.sh:15-23: 512 units, 8 heads, L = 30, M = 6)
    python bench.py --workload encode [--ids uniform]    # K1 alone at config 3 (HBM roofline)
    python bench.py --workload eval --gpus N     # row-sharded full-catalogue scoring + RCCL top-K all-gather (|I| = 20K, 1M)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HEADLINE = dict(num_items=20000, seqslen=100, num_units=128, num_heads=8, num_blocks=1, masklen=20,
                num_events=16, batch=512, time_scale=86400.0, learning_rate=5e-4, l2_reg=1e-4, ct_reg=1e-7,
                hidden_dropout_rate=0.1, attention_probs_dropout_rate=0.1)
# the published EasyDGL recipe (runme.sh:15-23 + the defaults of main.py:38,44): 512 units, 8 heads (head dim 64), 1 block,
# seqslen 30 (T = 31), masklen 6, batch 512, num_items 17771 — `--workload recipe`: the static engine with the unfused block tail
# (the fused one takes C in {64, 128}), the three-launch BiMAU of k_bimau_big.hip and the 16x16-tile scoring kernels at C = 512
RECIPE = dict(HEADLINE, num_items=17771, seqslen=30, num_units=512, masklen=6)

# the kernels whose launches in the timed region are bracketed with HIP events on their launch stream (library hook
# edgl_profile_next) — ONE kernel id per step, rotating through BRACKETS: an event record costs ~6 us of stream idle.  Id 0 is the
# dominant kernel: strip_kernel<ROLE_YF> — the row-side pass of the fused scoring / cross-entropy: ONE sweep over the item table
# computes the [R_w, I] logits, their row reference / sum AND the row gradients dl . table (edgl_score_flash_fwd_coef).  The
# scoring family carries 76 % of the step's algorithmic FLOPs.  With EDGL_SCORE_STRIP=0 the same slot times the
# round-2 kernel, with EDGL_FLASH_CE=0 the round-1 kernel (ROLE_Y: d_rows only, logits recomputed).
DOMINANT_KERNEL_ID = 0   # EDGL_KERNEL_SCORE_BWD_ROWS
DOMINANT_KERNEL = "strip_kernel<ROLE_YF> (bf16, C=128: forward logits + row reference / sum + d_rows = dl . table in one pass)"
DOMINANT_KERNEL_R2 = "score_bwd_kernel<ROLE_YF> (16x16 MFMA tiles, running row maxima: other widths / dtypes, EDGL_SCORE_STRIP=0)"
DOMINANT_KERNEL_R1 = "score_bwd_kernel<bf16, C/16=8, ROLE_Y> (d_rows = dl . table, logits recomputed)"


def flops_per_seq(c, rows_scored=None):
    """SURVEY.md §8d algorithmic FLOPs per sequence, forward; fwd+bwd = 3x.  `rows_scored`: weighted masked slots per
    sequence actually scored (default: all M — what the reference computes; label-0 slots have weight 0 and are skipped here)."""
    T, C, h, E, M, I, nb = c["seqslen"] + 1, c["num_units"], c["num_heads"], c["num_events"], c["masklen"], c["num_items"] + 1, c["num_blocks"]
    dh = C // h
    f = 0.0
    for i in range(nb):
        cin = 3 * C if i == 0 else C
        f += 2 * T * cin * 4 * C                  # F_qkvt
    f += nb * 6 * T * T * C                        # F_attn
    f += nb * 2 * T * C * E * (dh + 2)             # F_int
    f += nb * 2 * h * T * T * E                    # F_mark
    f += nb * 2 * T * C * C                        # F_proj
    f += nb * 8 * T * C * C                        # F_ffn
    f += 2 * T * C * C                             # F_head
    f += 2 * (M if rows_scored is None else rows_scored) * C * I     # F_score (train)
    return f


NBATCH = 8     # distinct device-resident batches rotated through the timed loop (the step never sees the same batch twice in a row)


def make_model_and_batch(c, dtype, device, seed, full_rows=False, nbatch=1, ids="zipf"):
    from types import SimpleNamespace
    import easydgl_amd
    from easydgl_amd import data as D
    F = SimpleNamespace(model="EasyDGL", num_items=c["num_items"], num_units=c["num_units"], num_heads=c["num_heads"],
                        num_blocks=c["num_blocks"], seqslen=c["seqslen"], masklen=c["masklen"], time_scale=c["time_scale"],
                        learning_rate=c["learning_rate"], l2_reg=c["l2_reg"], ct_reg=c["ct_reg"],
                        hidden_dropout_rate=c["hidden_dropout_rate"],
                        attention_probs_dropout_rate=c["attention_probs_dropout_rate"],
                        mark_table=D.synthetic_mark_table(c["num_items"], c["num_events"], multi_hot=bool(c.get("multi_hot", False))),
                        compute_dtype=dtype,
                        num_train_steps=None, num_warmup_steps=None, seed=9876)
    model = easydgl_amd.ranking(F).finalize(device)
    # full_rows: every sequence has all T tokens, so no masked slot falls on padding and every one of the B*M rows is scored
    batches = []
    for k in range(nbatch):
        sd = seed + 7919 * k
        ids_np, ts = D.synthetic_batch(c["num_items"], c["seqslen"], c["batch"], seed=sd, min_len=(c["seqslen"] + 1) if full_rows else 5,
                                       ids=ids)
        g = torch.Generator().manual_seed(sd)
        mp = D.draw_masked_positions(c["batch"], c["seqslen"] + 1, c["masklen"], generator=g)
        feats, labels = D.mask_random(torch.tensor(ids_np), torch.tensor(ts), c["num_items"], mp)
        batches.append(({k_: v.to(device).contiguous() for k_, v in feats.items()}, labels.to(device).contiguous()))
    if nbatch == 1:
        return model, batches[0][0], batches[0][1]
    return model, batches


def cpu_baseline(c, budget_s=15.0, nthreads=None, bs=None):
    """Restated reference graph on the host (TensorFlow is not installable offline): float32, reference op
    order incl. the materialised [hB,T,T(,E)] and [B*M,I] tensors, all host cores; bounded sample."""
    from oracle import easydgl_oracle as O
    from oracle import torch_ref as R
    # the reference pins intra/inter-op parallelism to 1 (src/main.py:167-168); on a many-core host the small
    # per-op tensors of this model do not scale past a few threads, so the baseline uses min(cores, 16) threads
    if nthreads is None:
        nthreads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(nthreads)
    bs = c["batch"] if bs is None else bs      # the GPU workload's own batch (512) unless a smaller sample is asked for
    cfg = O.Config(num_items=c["num_items"], seqslen=c["seqslen"], num_units=c["num_units"], num_heads=c["num_heads"],
                   num_blocks=c["num_blocks"], masklen=c["masklen"], time_scale=c["time_scale"], ct_reg=c["ct_reg"],
                   l2_reg=c["l2_reg"], learning_rate=c["learning_rate"], num_events=c["num_events"])
    rng = np.random.default_rng(9876)
    params = R.to_torch_params(O.init_params(cfg, rng), dtype=torch.float32)
    mt = O.synthetic_mark_table(cfg.num_items, cfg.num_events)
    ids, ts = O.synthetic_sequences(cfg, bs, rng)
    feats, labels = O.mask_random(cfg, ids, ts, O.draw_masked_positions(cfg, bs, rng))
    opt = R.TFAdam(params, cfg.learning_rate)
    R.cpu_train_step(cfg, params, opt, mt, feats, labels, torch.float32, 0.1, 0.1)   # warm-up
    t0, n = time.perf_counter(), 0
    while True:
        R.cpu_train_step(cfg, params, opt, mt, feats, labels, torch.float32, 0.1, 0.1)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 400:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n * bs / dt, 3), "unit": "sequences/s", "cores": nthreads, "kind": "port",
            "sample": f"{n} optimizer steps of batch {bs} (same shapes as the GPU workload), float32, "
                      f"oracle/torch_ref.py restatement of the TensorFlow graph (TF not installable offline)"}


def encode_bench(args):
    """K1 (EasyDGL.py:70-95) forward and backward alone at config 3: num_items 1M, seqslen 200 (T = 201), C = 256,
    B = 512, E = 16, bf16.  Algorithmic bytes per SURVEY §8d: ids+ts in, gathered item rows, X0 out, spans+marks out,
    position / mark tables.  One JSON line; `value` = forward GB/s."""
    from easydgl_amd import data as D
    from easydgl_amd import ops
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, T, C, E, num_items = 512, 201, 256, 16, 1_000_000
    I = num_items + 1
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    s = 2 if args.dtype == "bf16" else 4
    ids_mode = getattr(args, "ids", "zipf")
    ids_np, ts_np = D.synthetic_batch(num_items, T - 1, B, seed=9876, ids=ids_mode)
    ids, ts = torch.tensor(ids_np, device=dev), torch.tensor(ts_np, device=dev)
    unique_rows = int(np.unique(ids_np[ids_np != 0]).size)
    g = torch.Generator().manual_seed(1)
    item = (torch.randn((I, C), generator=g) * 0.02).to(dev).requires_grad_()
    pos = (torch.randn((T, C), generator=g) * 0.02).to(dev).requires_grad_()
    mk = (torch.randn((E, C), generator=g) * 0.02).to(dev).requires_grad_()
    item_c = item.detach().to(dt)
    mtab = torch.tensor(D.synthetic_mark_table(num_items, E), device=dev)
    tscale = ops.time_scales(C, dev) if hasattr(ops, "time_scales") else torch.pow(
        torch.tensor(10000.0), 2.0 * torch.arange(C // 2, dtype=torch.float32) / C).to(dev)
    drop = ops.NO_DROP

    def fwd():
        return ops.EncodeFn.apply(item, pos, mk, item_c, ids, ts, mtab, tscale, num_items, 86400.0, drop, dt)
    x0, _, _ = fwd()
    dx0 = torch.randn_like(x0)

    def timed(fn, n):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e-3

    t_f = timed(lambda: fwd(), args.steps)

    def bwd():
        x, _, _ = fwd()
        x.backward(dx0)
    t_fb = timed(bwd, args.steps)
    t_b = max(t_fb - t_f, 1e-9)
    bytes_f = B * T * (8 + 4) + B * T * C * s + B * T * 3 * C * s + B * T * (4 + E) + T * C * 4 + E * C * 4
    rows_touched = int((ids != 0).sum().item())
    bytes_b = B * T * 3 * C * s + B * T * (8 + E) + rows_touched * C * 4 * 2 + T * C * 4 + E * C * 4
    enc_traffic = None   # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/profile_encode.sh), read side x1
    for tp in (f"r03_encode_hbm_{ids_mode}.json", "r01_encode_hbm.json" if ids_mode == "zipf" else ""):
        tpath = os.path.join(ROOT, "profiles", tp)
        if tp and args.dtype == "bf16" and os.path.exists(tpath):
            enc_traffic = json.load(open(tpath)).get("hbm_bytes_per_launch_1x_read")
            break
    out = {"metric": "GB/s (K1 input encoding forward, |items|=1M L=200 d=256 B=512)", "value": round(bytes_f / t_f / 1e9, 1),
           "unit": "GB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(t_f * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "K1 encode (item gather x sqrt(C) + sinusoidal time code + position + mark embeddings), "
                                  f"config 3: num_items 1000000, seqslen 200 (T=201), num_units 256, batch 512, 16 marks, ids {ids_mode}",
                      "unique_item_rows": unique_rows, "algorithmic_bytes_fwd": bytes_f, "algorithmic_bytes_bwd": bytes_b},
           "roofline": {"bound": "hbm", "kernel": "encode_fwd_kernel", "achieved": round(bytes_f / t_f / 1e9, 1), "peak": 8000.0,
                        "unit": "GB/s", "frac": round(bytes_f / t_f / 8e12, 4), "traffic": enc_traffic},
           "backward": {"ms": round(t_b * 1e3, 4), "achieved_GBps": round(bytes_b / t_b / 1e9, 1),
                        "note": "encode_bwd + scatter into the touched rows of the [I, C] f32 table gradient (memset of the "
                                "full table gradient excluded from the byte count, included in the time)"}}
    return out


def baseline_model_bench(args):
    """Config 5 (and the CTSMA row f-4): one optimizer step (forward, backward, Adam) of TGAT / TiSASREC / CTSMA through the HIP
    attention kernels at the headline sizes B=512, seqslen 100, d=128, |items|=20K with the reference recipes' head / block
    counts (runme.sh:60-96).  A parity-case throughput, not the bench contract's headline line."""
    from types import SimpleNamespace
    import easydgl_amd
    from easydgl_amd import data as D
    from easydgl_amd._lib import profiler
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    name = {"tgat": "TGAT", "tisasrec": "TiSASREC", "ctsma": "CTSMA"}[args.workload]
    B, L, C, I = 512, 100, 128, 20000
    heads, blocks = {"TGAT": (1, 3), "TiSASREC": (8, 2), "CTSMA": (8, 2)}[name]
    F = SimpleNamespace(model=name, num_items=I, num_units=C, num_heads=heads, num_blocks=blocks, seqslen=L, timelen=256,
                        time_scale=86400.0, learning_rate=5e-4, l2_reg=1e-4, ct_reg=1e-7, hidden_dropout_rate=0.1,
                        attention_probs_dropout_rate=0.1, compute_dtype=args.dtype, num_train_steps=None, num_warmup_steps=None,
                        mark_table=D.synthetic_mark_table(I, 16))
    m = easydgl_amd.ranking(F).finalize(dev)
    ids_np, ts_np = D.synthetic_batch(I, L, B, seed=9876)           # records of L+1 tokens
    tok, tim = torch.tensor(ids_np, device=dev), torch.tensor(ts_np, device=dev)
    feats, labels = {"seqs_i": tok[:, :-1].contiguous(), "seqs_t": tim}, tok[:, 1:].contiguous()

    if args.path == "graph":   # the whole autograd step replayed as one HIP graph
        gstep = m.graphed_train_step(feats, labels)

        def step():
            return gstep(feats, labels)
    else:
        def step():
            return m.train_step(feats, labels)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.steps):
        loss = step()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.steps
    if hasattr(m, "check_inputs"):
        m.check_inputs()
    table = None
    if args.op_table:
        profiler.start()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        profiler.stop()
        table = {k: round(v[1] / 5, 4) for k, v in sorted(profiler.summary().items(), key=lambda kv: -kv[1][1])[:12]}
    out = {"metric": f"sequences/sec ({name} fwd+bwd+Adam) B=512 L=100 d=128 |I|=20K", "value": round(B / ms * 1e3, 1),
           "unit": "sequences/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"{name} optimizer step ({'autograd step replayed as a HIP graph' if args.path == 'graph' else 'autograd path'}), batch 512, seqslen 100, num_units 128, {heads} heads, "
                                  f"{blocks} blocks, num_items 20000, all-position loss, dropout 0.1/0.1, l2 1e-4"},
           "loss": round(float(loss), 5)}
    if table:
        out["ms_per_c_call"] = table
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)    # >= 20 warm-up + >= 100 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--path", default="engine", choices=["engine", "graph", "autograd"],
                    help="engine: static launch sequence issued eagerly (dominant kernel bracketed with HIP events); "
                         "graph: the same sequence replayed as one HIP graph; autograd: torch.autograd over the ops")
    ap.add_argument("--op-table", action="store_true", help="after the timed region, print a per-C-call time table to stderr")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary rows (M=6, all rows weighted, dropout off, "
                                                             "config-3 encode, sharded eval) the default N=1 run appends")
    ap.add_argument("--ids", default="zipf", choices=["zipf", "uniform"], help="item-id distribution of the synthetic sequences "
                    "(--workload encode; the step workload reports the uniform variant as an extra)")
    ap.add_argument("--workload", default="step", choices=["step", "recipe", "encode", "eval", "tgat", "tisasrec", "ctsma"],
                    help="step: the headline optimizer step (default, the bench contract); encode: K1 input encoding "
                         "(embedding gather + time code) alone at SURVEY §8d config 3 (|items| = 1M, L = 200, d = 256) — the "
                         "HBM-bound regime, reported as GB/s against the HBM roofline")
    args = ap.parse_args()
    if args.workload == "encode":
        return print(json.dumps(encode_bench(args)))
    if args.workload in ("tgat", "tisasrec", "ctsma"):
        return baseline_model_bench(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become N ranks (one per GPU) under torch.distributed.run
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.workload == "eval":
        return eval_bench(args, world, rank, local)
    # one rank per GPU; EDGL_BENCH_BACKEND=gloo lets the multi-process path be exercised on a single-GPU box (ranks then share
    # device 0 and the collective goes through the host) — a functional check only, never a measurement
    backend = os.environ.get("EDGL_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
    c = dict(RECIPE if args.workload == "recipe" else HEADLINE)
    from easydgl_amd import _lib
    if world > 1 and args.path == "autograd":
        # the loss normalises by sums over the GLOBAL batch (weighted rows, next-event marks: EasyDGL.py:183-185, temporal.py:331-333):
        # averaging per-rank gradients of per-rank losses is NOT the global-batch gradient (TrainEngine._global_counts
        # + a SUM all-reduce is) — the autograd path has no such protocol, so it does not pretend to be data parallel
        raise SystemExit("bench.py: --path autograd is single-GPU only (data-parallel steps run through the engine: --path engine | graph)")
    res = run_step_workload(c, args, dev, rank, world, dist, args.steps, args.warmup, bracket=(args.path != "graph"))
    dt, loss, dom = res["dt"], res["loss"], res["dom"]
    flash_engine = bool(getattr(res.get("engine"), "flash_ce", False))
    sync_loss = getattr(res.get("engine"), "sync_loss", None)
    # north_star's other multi-GPU path in the SAME line: the full-catalogue scoring step with the item table row-sharded over the
    # ranks and ONE packed RCCL all-gather of the local top-K.  Every rank takes part (collective); rank 0 reports.
    eval_sharded = vocab_parallel = None
    if world > 1 and args.workload == "step" and not args.no_extras:
        res.pop("engine", None)
        torch.cuda.empty_cache()
        eval_sharded = eval_rows(args, dev, world, rank, dist, steps=20, warmup=5, sizes=((20000, 128),))
        vocab_parallel = vocab_parallel_row(args, dev, world, rank, dist)

    if rank == 0:
        T, C, I, M = c["seqslen"] + 1, c["num_units"], c["num_items"] + 1, c["masklen"]
        R = c["batch"] * M
        h, E, nb = c["num_heads"], c["num_events"], c["num_blocks"]
        # rows with label 0 (masked slots that fell on padding) have weight 0 in the loss (EasyDGL.py:180) and are not
        # scored; only the weighted rows count as algorithmic work.  The timed loop rotates NBATCH batches: R_w is the mean over
        # the batches (whole step) / over the batches of the bracketed launches (dominant kernel).
        rows_w = res["rows_w"]
        R_w_mean = float(np.mean(rows_w))
        # ALGORITHMIC work of the dominant kernel.  Flash form: the kernel IS the forward scoring (logits
        # 2*R_w*C*I — F_score of the forward pass) and the row-gradient product d_rows = dl . table (2*R_w*C*I): both are
        # algorithmic, nothing is recomputed.  Round-1 form (EDGL_FLASH_CE=0): only the d_rows product is algorithmic, its
        # logits are a recomputation and count for `hw_util` (what the MFMA pipe did) alone.
        flash = flash_engine
        n_dom, ms_dom, bidx = dom[0]
        R_w = float(np.mean([rows_w[k] for k in bidx])) if bidx else R_w_mean
        dom_exec = 4.0 * R_w * C * I
        dom_flops = dom_exec if flash else 2.0 * R_w * C * I
        dom_ms = ms_dom / max(1, n_dom)
        peak = 2500.0 if args.dtype == "bf16" else 157.3
        # HBM bytes per launch of the dominant kernel: NOT measured by this run — PMC counters need their own rocprofv3 passes
        # (tools/refresh_profiles.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command); the committed summary of
        # the latest such pass is quoted and named in `traffic_source`
        traffic, traffic_source = None, None
        for tp in ("r06_dominant_kernel_traffic.json", "r05_dominant_kernel_traffic.json", "r04_dominant_kernel_traffic.json", "r03_dominant_kernel_traffic.json", "r02_dominant_kernel_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tp)
            if args.dtype == "bf16" and args.workload == "step" and os.path.exists(tpath):
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                traffic_source = f"profiles/{tp} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, read side x2 per the gfx950 guide; not this run)"
                break
        # whole-step HBM bytes from the same passes, and the time they would take at the 6.3 TB/s a streaming kernel reaches
        step_hbm = None
        for tp in ("r06_step_hbm_bytes.json", "r05_step_hbm_bytes.json", "r04_step_hbm_bytes.json"):
            tpath = os.path.join(ROOT, "profiles", tp)
            if step_hbm is None and args.dtype == "bf16" and args.workload == "step" and os.path.exists(tpath):
                sh = json.load(open(tpath))
                floor_ms = sh["total_bytes"] / 6.3e12 * 1e3
                step_hbm = {"bytes": sh["total_bytes"], "read_bytes_corrected": sh["read_bytes_corrected"], "write_bytes": sh["write_bytes"],
                            "floor_ms_at_6.3TBps": round(floor_ms, 4), "step_over_hbm_floor": round(dt / args.steps * 1e3 / floor_ms, 2),
                            "frac_of_hbm_floor": round(floor_ms / (dt / args.steps * 1e3), 4), "source": f"profiles/{tp} (not this run)"}
        ach = dom_flops / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        flops_done = 3 * flops_per_seq(c, rows_scored=R_w_mean / c["batch"]) * c["batch"]
        ms = res["step_ms"]
        if os.environ.get("EDGL_BENCH_DUMP_GROUPS"):
            print("step groups (ms):", [round(float(x), 3) for x in ms], file=sys.stderr)
        # the attention block (north_star: "MFMA utilisation for the attention/scoring blocks"): K3 forward (one launch) and the
        # three backward passes, bracketed like the dominant kernel; algorithmic FLOPs per SURVEY §8d: F_attn + F_int + F_mark
        # forward, twice that backward.  VALUBusy / MfmaUtil of the same kernels: profiles/r03_mfma_valu_util.txt.
        dh = C // h
        f_att_fwd = c["batch"] * nb * (6.0 * T * T * C + 2.0 * T * C * E * (dh + 2) + 2.0 * h * T * T * E)
        att = {}
        for key, kid, mult in (("forward", 2, 1.0), ("backward", 3, 2.0)):
            n_k, ms_k, _ = dom[kid]
            t_ms = ms_k / max(1, n_k) / max(1, nb)         # per BiMAU call (one per block)
            fl = mult * f_att_fwd / max(1, nb)
            att[key] = {"avg_ms": round(t_ms, 4), "algorithmic_gflop": round(fl / 1e9, 2),
                        "achieved": round(fl / (t_ms * 1e-3) / 1e12, 2) if t_ms > 0 else 0.0,
                        "frac": round(fl / (t_ms * 1e-3) / 1e12 / peak, 4) if t_ms > 0 else 0.0, "launches_timed": n_k}
        pu = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_mfma_valu_util.json") for r in (6, 5, 4, 3)) if os.path.exists(q)), "")
        # The roof these kernels actually sit under: the vector issue port of a SIMD (every VALU / transcendental / MFMA instruction of
        # a wave passes it: DESIGN.md rules 33, 35).  Floor = VALU instructions per launch (SQ_INSTS_VALU, separate PMC pass) x 3.1
        # cycles (measured issue cost of the cheapest class) / 1024 SIMDs / 2.1 GHz — a LOWER bound of the issue time (transcendentals
        # cost 9 cycles); `frac_of_issue_floor` = floor / measured launch time.  The MFMA fraction stays beside it because
        # north_star asks for it, not because the matrix pipe is what these kernels can fill.
        vi = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_valu_issue.json") for r in (6, 5)) if os.path.exists(q)), "")
        if os.path.exists(vi) and C == 128 and h == 8 and T == 101 and c["batch"] == 512 and E == 16 and nb == 1:
            vj = json.load(open(vi))
            fl_f = vj["bimau_fwd"]["issue_floor_us"] * 1e-3
            fl_b = sum(vj[k]["issue_floor_us"] for k in ("bimau_bwd_sweep1", "bimau_bwd_intensity", "bimau_bwd_sweep2")) * 1e-3
            for key, fl_ms, ks in (("forward", fl_f, ("bimau_fwd",)), ("backward", fl_b, ("bimau_bwd_sweep1", "bimau_bwd_intensity", "bimau_bwd_sweep2"))):
                att[key]["valu_insts_per_launch"] = sum(vj[k]["valu_insts_per_launch"] for k in ks)
                att[key]["issue_floor_ms"] = round(fl_ms, 4)
                att[key]["frac_of_issue_floor"] = round(fl_ms / att[key]["avg_ms"], 4) if att[key]["avg_ms"] > 0 else 0.0
                att[key]["mfma_frac"] = att[key]["frac"]
            att_bound = {"bound": "valu_issue", "issue_model": vj["_model"], "counter_source": os.path.relpath(vi, ROOT)}
        else:
            att_bound = {"bound": "valu_issue", "issue_model": None,
                         "note": "no SQ_INSTS_VALU table for this shape: only the MFMA fraction (`frac`) is reported"}
        out = {
            "metric": "sequences/sec (fwd+bwd+Adam) B=512 L=100 d=128 |I|=20K" if args.workload != "recipe" else
                      "sequences/sec (fwd+bwd+Adam) published recipe runme.sh:15-23: B=512 L=30 d=512 h=8 M=6 |I|=17.8K",
            "value": round(world * c["batch"] * args.steps / dt, 2),
            "unit": "sequences/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"EasyDGL optimizer step, per-GPU batch {c['batch']}, seqslen {c['seqslen']} (T={T}), num_units {C}, {h} heads, "
                                   f"{nb} block, num_items {c['num_items']} (I={I}), masklen {M}, {E} marks, dropout 0.1/0.1, ct_reg 1e-7, l2 1e-4",
                       "global_batch": world * c["batch"], "parallelism": f"dp{world}", "allreduce": res.get("allreduce"),
                       "padding": res.get("padding"),
                       "loss_mode": ("joined in front of every optimizer launch (TrainEngine.sync_loss = True)" if sync_loss else
                                     "deferred: the loss launches of step n are issued with step n+1, read once behind the timed steps "
                                     "(TrainEngine.sync_loss = False + join_loss(); train.py reads the loss at its logging points the same way)")
                                    if sync_loss is not None else "autograd path",
                       "batches_rotated": NBATCH,
                       "algorithmic_gflop_per_step_all_rows": round(3 * flops_per_seq(c) * c["batch"] / 1e9, 1),
                       "algorithmic_gflop_per_step_rows_scored": round(flops_done / 1e9, 1)},
            "loss": round(float(loss), 5), "path": args.path,
            "step_ms_hipevents": {"median": round(float(np.median(ms)), 4), "p10": round(float(np.percentile(ms, 10)), 4),
                                  "p90": round(float(np.percentile(ms, 90)), 4), "n": len(ms),
                                  "max": round(float(np.max(ms)), 4),
                                  "note": "mean step time of groups of 5 consecutive steps (one event record per group)"},
            "host_issue_ms_per_step": round(res["t_issue"] / args.steps * 1e3, 4),
            "roofline": {"bound": "mfma", "kernel": (DOMINANT_KERNEL if C == 128 and args.dtype == "bf16" and os.environ.get("EDGL_SCORE_STRIP", "1") != "0"
                                                      else DOMINANT_KERNEL_R2) if flash else DOMINANT_KERNEL_R1,
                         "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         "hw_util": round(dom_exec / (dom_ms * 1e-3) / 1e12 / peak, 4) if dom_ms > 0 else 0.0,
                         "avg_launch_ms": round(dom_ms, 4), "algorithmic_flop": dom_flops, "executed_flop": dom_exec,
                         "rows_scored": round(R_w, 1), "rows_total": R, "traffic": traffic, "traffic_source": traffic_source,
                         "launches_timed": n_dom},
            "roofline_attention": {**att_bound, "mfma_peak": peak, "peak": peak, "unit": "TFLOP/s (frac / mfma_frac: against the dense MFMA peak); ms (issue_floor_ms)",
                                   "kernels": "K3 BiMAU: bimau_fwd_kernel | bimau_bwd_sweep1 + intensity_bwd + bimau_bwd_sweep2 "
                                              "(QK^T, softmax, P.T_, intensity MLP, lambda.marks^T, (G.P).V and their backward)",
                                   "forward": att["forward"], "backward": att["backward"],
                                   "pipe_utilisation_pmc": json.load(open(pu)) if os.path.exists(pu) else None},
            # whole-step MFMA fraction on the work actually done (weight-0 rows are skipped exactly, so they are not counted)
            "whole_step_mfma_frac": round(flops_done / (dt / args.steps) / 1e12 / peak, 4),
            "step_hbm_bytes": step_hbm,
            # what the element-wise parity of this mode rests on: the benchmarked step runs dropout 0.1 / 0.1, whose masks no
            # oracle can reproduce — tensors are compared with dropout off, the dropout itself statistically
            "parity_note": "element-wise parity vs the oracle: dropout off (tests/test_gpu_headline_parity.py at this size); dropout on: "
                           "unbiasedness over 1600 masks, finite-difference backward under a fixed mask, stored keep bits == hashed "
                           "masks bit for bit (tests/test_gpu_coding.py, test_gpu_ops.py)",
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(c, budget_s=12.0)      # the GPU workload's own batch of 512
            # the reference's own thread setting (src/main.py:167-168: intra/inter-op parallelism = 1), on a batch of 16
            out["cpu_baseline_1thread_batch16"] = cpu_baseline(c, budget_s=8.0, nthreads=1, bs=16)
        if world == 1 and not args.no_extras and args.path == "engine":
            out["extras"] = extras(c, args, dev)
        if eval_sharded is not None:
            out["extras"] = {"eval_sharded": eval_sharded, "vocab_parallel_ce": vocab_parallel}
        if args.op_table:
            step = res["step"]
            _lib.profiler.start()
            for i_ in range(5):
                step(i_)
            torch.cuda.synchronize()
            _lib.profiler.stop()
            rows = sorted(_lib.profiler.summary().items(), key=lambda kv: -kv[1][1])
            tot = sum(v[1] for _, v in rows)
            print("per-call GPU time over 5 steps (HIP events):", file=sys.stderr)
            for k, (n, ms_) in rows:
                print(f"  {k:28s} calls {n:4d}  {ms_ / 5:9.3f} ms/step  {100 * ms_ / tot:5.1f}%", file=sys.stderr)
            print(f"  total {tot / 5:.3f} ms/step", file=sys.stderr)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


BRACKETS = (0, 2, 3)   # EDGL_KERNEL_SCORE_BWD_ROWS, EDGL_KERNEL_BIMAU_FWD, EDGL_KERNEL_BIMAU_BWD_ALL (include/easydgl_hip.h)


def run_step_workload(c, args, dev, rank, world, dist, steps, warmup, bracket=False, full_rows=False, ids="zipf", device_masker=False):
    """W warm-up steps, then EXACTLY `steps` optimizer steps between barrier + synchronize on both sides; the time is the max
    over ranks.  The steps rotate through NBATCH distinct batches resident in HBM (the engine is pointed at them: no copies);
    `device_masker`: every step first draws its masked positions on the device (edgl_mask_random, row a-1) from the unmasked
    sequences.  Groups of steps are bracketed by HIP events on the launch stream (median / p10 / p90)."""
    from easydgl_amd import _lib, parallel
    from easydgl_amd import data as D
    model, batches = make_model_and_batch(c, args.dtype, dev, seed=9876 + rank, full_rows=full_rows, nbatch=NBATCH, ids=ids)
    raw = None
    if device_masker:      # unmasked sequences of the same batches; the masker writes fresh features / labels every step
        raw = []
        for k in range(NBATCH):
            ids_np, ts = D.synthetic_batch(c["num_items"], c["seqslen"], c["batch"], seed=9876 + rank + 7919 * k,
                                           min_len=(c["seqslen"] + 1) if full_rows else 5, ids=ids)
            raw.append((torch.tensor(ids_np, device=dev), torch.tensor(ts, device=dev)))
    eng = None
    if args.path == "autograd":
        def step(i=0):
            from easydgl_amd import ops
            feats, labels = batches[i % NBATCH]
            ops.rng_advance(model._rng_state)
            model.zero_grad_arena()
            loss = model.train_loss(feats, labels)
            loss.backward()
            model.optimizer_step()
            return loss.detach()
    else:
        from easydgl_amd.engine import TrainEngine
        eng = TrainEngine(model, c["batch"], use_graph=(args.path == "graph"), process_group=None)
        eng.load_batch(*batches[0])
        # the loss of a step is read once, behind the timed region's device-wide synchronisation — as a training loop reads it every
        # few hundred steps (train.py): the engine then leaves its loss kernels on the side stream instead of joining them in
        # front of every Adam launch (TrainEngine.sync_loss / join_loss; EDGL_BENCH_SYNC_LOSS=1: the joined form)
        eng.sync_loss = os.environ.get("EDGL_BENCH_SYNC_LOSS", "0") == "1"

        def step(i=0):
            if raw is not None:
                feats, labels = D.device_mask_random(raw[i % NBATCH][0], raw[i % NBATCH][1], model.mask, c["masklen"], model._rng_state,
                                                     stream_id=0x4d41534b + i)
            else:
                feats, labels = batches[i % NBATCH]
            if args.path == "graph":
                return eng.step(feats, labels)        # the captured launches read the static buffers: four small copies
            eng.bind_batch(feats, labels)
            return eng.step()

    # Everything the timed region needs exists BEFORE the warmup steps (HIP event handles exist after a first record): nothing but the
    # prescribed synchronisation sits between the warmup and the timed steps.  (After an idle stretch of milliseconds — event creation
    # or a garbage collection in that gap — the first ~30 steps run up to 17 % slower, eager and graph path alike: step groups of one
    # run 0.953, 0.875, 0.848, 0.833, 0.825, 0.819, 0.815 ... ms.  `value` includes whatever ramp is left; `step_ms_hipevents.median`
    # is the steady state.)
    evs = []
    for _ in range(steps):
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b_.record()
        evs.append((a, b_))
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    for e in marks:
        e.record()
    # The eager path issues ~33 launches per step from Python: a garbage-collector pass in the middle of the timed steps (generation
    # 2 walks every object torch and numpy keep alive) stalls the launch stream for milliseconds.  Collected here — BEFORE the warmup,
    # so that the GPU does not sit idle between the warmup and the timed steps — and disabled until the timed steps are done.
    import gc
    gc.collect()
    gc.disable()
    # Device conditioning (not optimizer steps): the ramp above starts from wherever the device was — after the model build it is idle,
    # and W = 5 warmup steps (4 ms) leave the whole of a K = 20 run inside it (0.903, 0.868, 0.855, 0.839 ms per step group against
    # 0.813 steady).  Our own tiled GEMM runs on scratch operands for EDGL_BENCH_SPIN_MS (default 60 ms; 0 disables) right in front
    # of the W warmup steps, so that they — and the K timed steps — see the clocks a training job sees after its first second.
    spin_ms = float(os.environ.get("EDGL_BENCH_SPIN_MS", "60"))
    if spin_ms > 0 and args.dtype == "bf16":
        from easydgl_amd import ops as _ops
        sa = torch.randn(16384, 512, device=dev).bfloat16()
        sw = torch.randn(512, 512, device=dev).bfloat16()
        so = torch.empty(16384, 512, device=dev, dtype=torch.bfloat16)
        ts = time.perf_counter()
        while (time.perf_counter() - ts) * 1e3 < spin_ms:
            for _ in range(64):
                _ops.gemm(sa, sw, 16384, 512, 512, 512, 512, True, False, torch.bfloat16, out=so)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # HIP events inside the timed region cost ~6 us of launch-stream idle each (measured: kernel timeline): ONE kernel group is
    # bracketed per step, three steps out of BR_EVERY = 8 or 16 (scoring rows pass, BiMAU forward, BiMAU backward), and the step marks are
    # recorded every MARK_EVERY steps
    # (4: the brackets cost 1.2 % of the step; 8: 0.6 %; 16 from 64 steps on: 0.3 %, still >= 4 launches per bracketed kernel group)
    BR_EVERY = max(4, int(os.environ.get("EDGL_BENCH_BRACKET_EVERY", "16" if steps >= 64 else "8")))
    MARK_EVERY = int(os.environ.get("EDGL_BENCH_MARK_EVERY", "5"))
    br_used = {k: [] for k in BRACKETS}
    for i in range(steps):
        if i % MARK_EVERY == 0:
            marks[i].record()
        if bracket and i % BR_EVERY < len(BRACKETS):
            kid = BRACKETS[i % BR_EVERY]
            _lib.lib.edgl_profile_next(kid, evs[i][0].cuda_event, evs[i][1].cuda_event)
            br_used[kid].append(i)
        loss = step(warmup + i)
    if eng is not None:
        eng.join_loss()      # (sync_loss False: the last step's loss launches, inside the timed region)
    marks[steps].record()
    t_issue = time.perf_counter() - t0      # host time to ISSUE the steps (the GPU is still running: launch-bound iff this ~ dt)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    _lib.lib.edgl_profile_next(-1, None, None)
    # (count, total ms, batch index of every bracketed step) per kernel group
    dom = {k: (len(v), sum(evs[i][0].elapsed_time(evs[i][1]) for i in v), [(warmup + i) % NBATCH for i in v]) for k, v in br_used.items()}
    mk = list(range(0, steps, MARK_EVERY)) + [steps]
    step_ms = [marks[a].elapsed_time(marks[b_]) / (b_ - a) for a, b_ in zip(mk[:-1], mk[1:])]   # mean step time per group
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if not np.isfinite(float(loss)):
        raise RuntimeError("loss is not finite")
    rows_w = [int((lb != 0).sum().item()) for _, lb in batches]
    # what padding looks like to the attention kernels in these batches (DESIGN §4.6 rule 50): positions with id 0, and 16-key tiles
    # in front of a sequence's first non-zero id (the only ones BiMAU can leave out: a MASK token on a padded position is a real key)
    pad_pos, pad_tiles, n_pos, n_tiles = 0, 0, 0, 0
    for f, _ in batches:
        ids = f["seqs_i"]
        nz = ids != 0
        first = torch.where(nz.any(dim=1), nz.to(torch.int64).argmax(dim=1), torch.full((ids.shape[0],), ids.shape[1], device=ids.device))
        pad_pos += int((~nz).sum().item()); n_pos += ids.numel()
        pad_tiles += int((first // 16).sum().item()); n_tiles += ids.shape[0] * ((ids.shape[1] + 15) // 16)
    # N > 1: what the step's ONE collective costs where it stands — the SUM all-reduce of the flat f32 gradient arena between the last
    # backward kernel and the optimizer, nothing beside it (the item table's gradient receives the embedding scatter's rows in the last
    # kernel of the backward: DESIGN.md §6) — HIP events around it on 6 further steps OUTSIDE the timed region
    ar = None
    if world > 1 and eng is not None:
        eng._ar_events = []
        for i in range(6):
            step(warmup + steps + i)
        eng.join_loss()
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b_) for a, b_ in eng._ar_events][1:]
        eng._ar_events = None
        ar = {"allreduce_bytes_per_rank": int(model._grad_comm.numel() * 4), "allreduce_exposed_ms": round(float(np.mean(ms)), 4) if ms else None,
              "allreduce_exposed_share_of_step": round(float(np.mean(ms)) / (dt / steps * 1e3), 4) if ms else None,
              "note": "one SUM all-reduce per step, fully exposed (measured with HIP events on 5 steps behind the timed region); "
                      "EDGL_BENCH_BACKEND=gloo stages it through the host"}
    padding = {"positions_with_id_0": round(pad_pos / max(1, n_pos), 4), "key_tiles_bimau_can_skip": round(pad_tiles / max(1, n_tiles), 4),
               "note": "the reference's masker draws masked positions over padding too (dataloader.py:187-191): MASK tokens are real keys"}
    return {"padding": padding, "allreduce": ar, "dt": dt, "t_issue": t_issue, "loss": loss, "rows_w": rows_w, "dom": dom, "step_ms": step_ms, "step": step, "model": model,
            "engine": None if args.path == "autograd" else eng}


def extras(c, args, dev):
    """Secondary rows of SURVEY §8(d), measured after the headline line on the same GPU (N = 1 only): the reference's default
    masklen 6, a batch whose masked slots all carry weight, dropout off, the multi-hot mark table, the config-3 K1 line and the
    sharded-eval step.  Short runs (30 timed steps each); none of them is `value`."""
    import copy
    out = {}
    a2 = copy.copy(args)

    def row(cc, full_rows=False, multi_hot=False, ids="zipf", device_masker=False, dtype=None, steps=30, warmup=10, bracket=False, path=None):
        cc = dict(cc, multi_hot=multi_hot)
        ar = a2
        if dtype is not None:
            ar = copy.copy(a2)
            ar.dtype = dtype
        if path is not None:
            ar = copy.copy(ar)
            ar.path = path
        r = run_step_workload(cc, ar, dev, 0, 1, None, steps, warmup, bracket=bracket, full_rows=full_rows, ids=ids, device_masker=device_masker)
        rw = float(np.mean(r["rows_w"]))
        fl = 3 * flops_per_seq(cc, rows_scored=rw / cc["batch"]) * cc["batch"]
        ms = float(np.median(r["step_ms"]))
        out_row = {"ms_per_step": round(r["dt"] / steps * 1e3, 4), "ms_median": round(ms, 4),
                   "sequences_per_s": round(cc["batch"] * steps / r["dt"], 1), "rows_scored": round(rw, 1), "rows_total": cc["batch"] * cc["masklen"],
                   "whole_step_mfma_frac": round(fl / (r["dt"] / steps) / 1e12 / (2500.0 if ar.dtype == "bf16" else 157.3), 4)}
        d0 = r["dom"].get(0) if isinstance(r.get("dom"), dict) else None
        if d0 and d0[0] > 0 and d0[1] > 0 and ar.dtype == "bf16":
            # the row pass of the scoring (one launch = 2 products x 2 R_w C I, measured live with HIP events on its stream)
            dms = d0[1] / d0[0]
            out_row["scoring_row_pass_ms"] = round(dms, 4)
            out_row["scoring_row_pass_mfma_frac"] = round(4.0 * rw * cc["num_units"] * (cc["num_items"] + 1) / (dms * 1e-3) / 2.5e15, 4)
        return out_row
    out["masklen_6"] = row(dict(c, masklen=6))
    out["all_rows_weighted"] = row(c, full_rows=True)
    out["dropout_off"] = row(dict(c, hidden_dropout_rate=0.0, attention_probs_dropout_rate=0.0))
    out["multi_hot_marks"] = row(c, multi_hot=True)
    # ids ~ U[1, num_items): no clip pile-up on one id (the Zipf recipe of SURVEY §8d puts ~35 % of the tokens and labels on id
    # num_items - 1, which flatters the gather, the embedding scatter and the label paths)
    out["uniform_ids"] = row(c, ids="uniform")
    # row a-1 inside the step: the masked positions of every batch are drawn on the device (edgl_mask_random) right before it
    out["with_device_masker"] = row(c, device_masker=True)
    # the reference's own arithmetic: float32 activations and weights end to end (exact-f32 MFMA v_mfma_f32_16x16x4_f32: the parity
    # mode of every kernel, 157 TFLOP/s peak) — the same engine, launch sequence and step; the headline runs bf16 as BASELINE.json asks
    if args.dtype == "bf16":
        out["f32_reference_arithmetic"] = dict(row(c, dtype="f32"), dtype="f32",
                                               note="whole_step_mfma_frac against the 157.3 TFLOP/s dense f32 MFMA peak")
    # the published recipe's shape (runme.sh:15-23) through the same engine: `python bench.py --workload recipe` prints it as a line
    out["recipe_runme_sh"] = dict(row(dict(RECIPE)), workload="num_units 512, 8 heads, seqslen 30, masklen 6, batch 512, num_items 17771")
    # the reference's DEFAULT flags (main.py:35-38,44,60-66): 50 units in ONE head, 3 blocks, seqslen 30, masklen 6, batch 128, no
    # dropout / regularisers — head dim 50 runs zero-padded to 64 channels, i.e. on the head-dim-64 kernels
    cdef = dict(c, num_units=50, num_heads=1, num_blocks=3, seqslen=30, masklen=6, batch=128, l2_reg=0.0, ct_reg=0.0,
                hidden_dropout_rate=0.0, attention_probs_dropout_rate=0.0)
    out["reference_default_flags"] = dict(row(cdef), workload="num_units 50, 1 head, 3 blocks, seqslen 30, masklen 6, batch 128, num_items 20000",
                                          path="engine (eager issue)")
    # the same step replayed as ONE HIP graph (TrainEngine(use_graph=True)): at batch 128 the step is launch bound, where the replay wins
    # (at the headline shape the eager issue does: DESIGN.md rule 57)
    try:
        out["reference_default_flags_graph"] = dict(row(cdef, path="graph"), path="graph (HIP-graph replay of the same launch sequence)")
    except Exception as e:      # (a capture failure must not cost the line)
        out["reference_default_flags_graph"] = {"error": repr(e)[:200]}
    torch.cuda.empty_cache()
    a3 = copy.copy(args)
    a3.steps, a3.warmup = 50, 10
    for key, ids_mode in (("encode_config3", "zipf"), ("encode_config3_uniform_ids", "uniform")):
        a3.ids = ids_mode
        enc = encode_bench(a3)
        tr = enc["roofline"]["traffic"]
        out[key] = {"algorithmic_GBps": enc["value"], "ms": enc["ms_per_step"], "frac_of_8TBps_algorithmic": enc["roofline"]["frac"],
                    "hbm_side_bytes_per_launch_pmc": tr,
                    "hbm_side_GBps_pmc": round(tr / (enc["ms_per_step"] * 1e-3) / 1e9, 1) if tr else None,
                    # counter bytes below the algorithmic bytes = gathered rows served from L2 / MALL: the honest HBM fraction
                    "frac_of_8TBps_counter_side": round(tr / (enc["ms_per_step"] * 1e-3) / 8e12, 4) if tr else None,
                    "unique_item_rows": enc["config"]["unique_item_rows"], "workload": enc["config"]["workload"]}
    torch.cuda.empty_cache()
    # BASELINE.json configs[2] as a whole optimizer step (|items| = 1 M, L = 200 -> T = 201, d = 256, 8 heads, masklen 40, batch 512,
    # bf16; the K1 line above is the HBM-bound kernel of this config): engine path, the unfused block tail and the wide strip scoring
    # kernels of k_score_stripw.hip (DESIGN.md §4.7 rules 65-67).  Short: 3 + 8 steps of ~26 ms.
    if args.dtype == "bf16":
        c3 = dict(c, num_items=1_000_000, seqslen=200, num_units=256, masklen=40)
        out["config3_step"] = dict(row(c3, steps=8, warmup=3, bracket=True),
                                   workload="EasyDGL optimizer step at BASELINE.json configs[2]: num_items 1000000 (I = 1000001), seqslen 200 "
                                            "(T = 201), num_units 256, 8 heads, 1 block, masklen 40, batch 512")
        torch.cuda.empty_cache()
    out["eval_sharded"] = eval_rows(args, dev, 1, 0, None, steps=20, warmup=5, sizes=((20000, 128), (1_000_000, 256)))
    out["vocab_parallel_ce"] = vocab_parallel_row(args, dev, 1, 0, None)
    return out


def _lib_supported(R, C, n, T, K):
    from easydgl_amd import _lib
    return _lib.lib.edgl_score_topk_fused_supported(R, C, n, T, K, _lib.BF16)


def vocab_parallel_row(args, dev, world, rank, dist, steps=20, warmup=5):
    """SURVEY §8e row 3 as a timed row: the headline's scoring + cross-entropy step (10240 masked slots of 128 channels, 47.5 % of them
    without weight, 20001 items; EasyDGL.py:149-155,177-185) with the item table ROW-SHARDED over the ranks — every rank scores the same
    rows against its shard with the strip kernels (ops.vocab_parallel_ce), ONE packed all-gather of (log-sum-exp, label logit), ONE
    all-reduce of d_rows.  The rows are shared, so the job does not grow with N: strong scaling.  Every rank takes part; errors are
    reported in the row instead of costing the line."""
    try:
        from easydgl_amd import ops
        g = torch.Generator(device="cpu").manual_seed(4242)
        R, C, I = 10240, 128, 20001
        dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        rows = (torch.randn((R, C), generator=g) * 0.6).to(dt).to(dev)
        table = (torch.randn((I, C), generator=g) * 0.4).to(dt).to(dev)
        bias = (torch.randn(I - 1, generator=g) * 0.3).to(dev)
        labels = torch.randint(1, I, (R,), generator=g)
        labels[torch.rand(R, generator=g) > 0.525] = 0
        labels = labels.to(dev)

        def call():
            return ops.vocab_parallel_ce(rows, table, bias, labels)
        for _ in range(warmup):
            loss = call()[0]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = call()[0]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dtm], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtm = float(t.item())
        return {"rows": R, "weighted_rows": int((labels != 0).sum()), "num_units": C, "num_items": I - 1, "shards": world,
                "ms_per_call": round(dtm / steps * 1e3, 4), "loss": round(float(loss), 5), "scaling": "strong",
                "collectives": "one all-gather of [R, 2] f32 + one all-reduce of d_rows [R, C] f32 per call",
                "note": "loss + d_rows + the shard's d_table / d_bias; host-driven torch ops between the launches included"}
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)[:300]}


def eval_rows(args, dev, world, rank, dist, steps, warmup, sizes):
    """Sequential.eval's scoring step (Base.py:150-181) with the item table row-sharded over the ranks: every
    rank holds the SAME evaluation batch of 512 sequences (the encoder is replicated), scores it against its table shard,
    masks the seen ids that fall in the shard, keeps a local top-100 with global ids, and ONE packed all-gather + the merge
    kernel give the global top-100.  The batch is shared, so the job does not grow with N: strong scaling."""
    from types import SimpleNamespace
    import easydgl_amd
    from easydgl_amd import data as D
    rows = []
    for num_items, C in sizes:
        cfgd = dict(HEADLINE, num_items=num_items, num_units=C, seqslen=100 if C == 128 else 200, masklen=20 if C == 128 else 40)
        F = SimpleNamespace(model="EasyDGL", num_items=num_items, num_units=C, num_heads=8, num_blocks=1, seqslen=cfgd["seqslen"],
                            masklen=cfgd["masklen"], time_scale=86400.0, learning_rate=5e-4, l2_reg=1e-4, ct_reg=1e-7,
                            hidden_dropout_rate=0.1, attention_probs_dropout_rate=0.1,
                            mark_table=D.synthetic_mark_table(num_items, 16), compute_dtype=args.dtype, num_train_steps=None,
                            num_warmup_steps=None, seed=9876)
        model = easydgl_amd.ranking(F).finalize(dev)
        ids, ts = D.synthetic_batch(num_items, cfgd["seqslen"], 512, seed=9876)     # the same batch on every rank
        feats, _ = D.device_mask_last(torch.tensor(ids, device=dev), torch.tensor(ts, device=dev), model.mask)
        K = 100

        def step():
            return model.eval_topk_sharded(feats, mask_seen=True, K=K)
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            val, idx = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ag_ms = None
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            # the collective alone: one packed [512, 2K] 32-bit buffer per rank
            packed = torch.zeros((512, 2 * K), device=dev, dtype=torch.int32)
            flat = torch.empty((world * 512, 2 * K), device=dev, dtype=torch.int32)
            for _ in range(5):
                dist.all_gather_into_tensor(flat, packed)
            torch.cuda.synchronize()
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                dist.all_gather_into_tensor(flat, packed)
            b_.record()
            torch.cuda.synchronize()
            ag_ms = a.elapsed_time(b_) / 20
        assert idx.shape == (512, K) and int(idx.min()) >= 0
        # roofline statement of the evaluation step's own kernel K6 (seen mask + top-K of a [512, n] f32 logits tile that the scoring
        # GEMM wrote just before): algorithmic bytes = ONE read of the tile, rows x n x 4 B, over the launch time, against the 8 TB/s
        # HBM peak.  Rows of up to 20 472 items run the register form (mask_topk_reg_kernel: the row is read once, candidates above
        # the K-th largest thread maximum are ranked in LDS); longer chunks the four-pass radix select, which sweeps the row five times
        # (served by L2 / MALL): `executed_passes`.
        from easydgl_amd import ops as _o
        i0s, i1s = (0, model.num_items) if world == 1 else __import__("easydgl_amd").parallel.shard_bounds(model.num_items, world, rank)
        nloc = max(1024, (_o.EVAL_TILE_BYTES // (4 * 512)) // 8 * 8)
        if K <= 128 and nloc > _o.TOPK_REG_ITEMS >= 1024:      # (the chunk rule of ops.score_topk)
            nloc = _o.TOPK_REG_ITEMS
        nloc = min(i1s - i0s, nloc)
        lg = torch.randn((512, nloc), device=dev, dtype=torch.float32)
        for _ in range(3):
            _o.mask_topk(lg, i0s, feats["seqs_i"], K)
        torch.cuda.synchronize()
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        for _ in range(10):
            _o.mask_topk(lg, i0s, feats["seqs_i"], K)
        eb.record()
        torch.cuda.synchronize()
        k6_ms = ea.elapsed_time(eb) / 10
        k6_bytes = 512 * nloc * 4
        k6_reg = nloc <= 256 * 80 - 8 and os.environ.get("EDGL_TOPK_REG", "1") != "0"
        del lg
        # the fused form (csrc/k_eval_topk.hip: two sweeps of rows . table^T on the matrix pipe, candidates above a per-row bound, one
        # ranking launch — no logits tile): time of the whole op on this rank's shard, ALGORITHMIC FLOPs = one logits computation
        # 2 R C n (the second sweep is a recomputation: `executed_flop`), against the dense bf16 MFMA peak
        fused = None
        if _o.EVAL_FUSED and args.dtype == "bf16":
            rws = torch.randn((512, C), device=dev).bfloat16()
            tabc = model.compute(model.item_embs.lookup_table)
            for _ in range(3):
                _o.score_topk(rws, tabc, model.output_bias, feats["seqs_i"], K, i0s, i1s)
            torch.cuda.synchronize()
            ea.record()
            for _ in range(10):
                _o.score_topk(rws, tabc, model.output_bias, feats["seqs_i"], K, i0s, i1s)
            eb.record()
            torch.cuda.synchronize()
            f_ms = ea.elapsed_time(eb) / 10
            f_alg = 2.0 * 512 * C * (i1s - i0s)
            taken = bool(_lib_supported(512, C, min(i1s - i0s, 131072 if C == 256 else 262144), feats["seqs_i"].shape[1], K))
            fused = {"bound": "mfma", "kernel": "eval_sweep_kernel x 2 (group maxima | candidates) + eval_thr_kernel + eval_rank_kernel: scoring + seen "
                                                "mask + top-K without a logits tile" if taken else "not taken at this shape: logits tile + mask_topk",
                     "avg_op_ms": round(f_ms, 4), "algorithmic_flop": f_alg, "executed_flop": 2 * f_alg if taken else f_alg,
                     "achieved": round(f_alg / (f_ms * 1e-3) / 1e12, 2), "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": round(f_alg / (f_ms * 1e-3) / 1e12 / 2500.0, 4),
                     "hbm_bytes_not_written": 512 * (i1s - i0s) * 4}
            del rws
        rows.append({"num_items": num_items, "num_units": C, "T": cfgd["seqslen"] + 1, "batch": 512, "K": K, "shards": world,
                     "ms_per_eval_step": round(dt / steps * 1e3, 4), "sequences_per_s": round(512 * steps / dt, 1),
                     # (with the fused evaluation scoring on the path — `roofline_scoring_fused.kernel` says whether it was taken — this
                     #  kernel does NOT run in the step above: it is the unfused fallback of shapes the fused form declines, timed alone)
                     "roofline_k6": {"bound": "hbm", "on_eval_path": not (fused is not None and fused["kernel"].startswith("eval_sweep")),
                                     "kernel": ("UNFUSED FALLBACK, timed alone — " if (fused is not None and fused["kernel"].startswith("eval_sweep")) else "") +
                                               ("mask_topk_reg_kernel (seen mask, row in registers, candidates ranked in LDS)" if k6_reg else
                                                "mask_topk_kernel (seen mask + 4-pass radix select + tie pass)") + ", one logits chunk",
                                     "logits_per_row": nloc, "executed_passes": 1 if k6_reg else 5, "algorithmic_bytes": k6_bytes,
                                     "avg_launch_ms": round(k6_ms, 4),
                                     "achieved": round(k6_bytes / (k6_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                     "frac": round(k6_bytes / (k6_ms * 1e-3) / 8e12, 4)},
                     "roofline_scoring_fused": fused,
                     "allgather_bytes_per_rank": 512 * 2 * K * 4, "allgather_bytes_gathered": world * 512 * 2 * K * 4,
                     "allgather_ms": None if ag_ms is None else round(ag_ms, 4)})
        del model
        torch.cuda.empty_cache()
    return rows


def eval_bench(args, world, rank, local):
    """--workload eval: the row-sharded full-catalogue scoring step at |I| = 20K (headline) and |I| = 1M (config 3 sizes)."""
    backend = os.environ.get("EDGL_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    sizes = ((20000, 128), (1_000_000, 256))
    if os.environ.get("EDGL_BENCH_EVAL_ROWS"):     # profiling: only the first n catalogue sizes (tools/ktrace.sh --workload eval)
        sizes = sizes[:int(os.environ["EDGL_BENCH_EVAL_ROWS"])]
    rows = eval_rows(args, dev, world, rank, dist, args.steps, args.warmup, sizes=sizes)
    if rank == 0:
        head = rows[0]
        print(json.dumps({"metric": "sequences/sec (evaluation: encode + sharded full-catalogue scoring + seen mask + top-100) B=512 L=100 d=128 |I|=20K",
                          "value": head["sequences_per_s"], "unit": "sequences/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": head["ms_per_eval_step"], "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                          "config": {"workload": "Sequential.eval scoring step, item table row-sharded over the ranks, one packed "
                                                 "RCCL all-gather of the local top-100 + merge kernel", "parallelism": f"shard{world}"},
                          "rows": rows}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
