"""Benchmark of the batch hot path on MI355X.

One GPU (the default): one step = one pass of the alias matcher (K1) over one batch of
synthetic modalias strings already resident in HBM - configs[1] of the benchmark definition,
1000 strings x 30000 alias patterns, seed 2.  Beside the headline value the line carries
  roofline      algorithmic bytes of one K1 launch / measured device time
  cpu_baseline  the CPU oracle (glibc fnmatch loop) timed on the host cores
  lists         match lists against first / count alone, 70000 strings on the same table
  k2            dependency flags + closure on configs[2] (1M modules, ~10M edges)
  fleet         the fused inventory pipeline on configs[4] (100000 hosts, ~10M devices)
  configs3      1M strings x 3M patterns on this one GPU

--gpus N > 1 (one rank per GPU: launched through torch.distributed.run, or by this script itself
as child processes when it is called without a launcher): configs[3], the alias
table cut into N slices, all strings against every slice, MIN / SUM all-reduce over RCCL per
step; the collective alone, the string-sharded layout and the one-GPU time are in the same line.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--strings", type=int, default=1000, help="strings per GPU per step")
    ap.add_argument("--patterns", type=int, default=30000)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--graph-vertices", type=int, default=1_000_000)
    ap.add_argument("--graph-degree", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=31,
                    help="one GPU: how many brackets of --steps steps are timed; the median one is reported")
    ap.add_argument("--c3-strings", type=int, default=1_000_000, help="configs[3]: strings")
    ap.add_argument("--c3-patterns", type=int, default=3_000_000, help="configs[3]: alias patterns")
    ap.add_argument("--c3-cpu-strings", type=int, default=128, help="configs[3]: strings of the CPU sample")
    ap.add_argument("--no-configs3", action="store_true")
    ap.add_argument("--no-k2", action="store_true")
    ap.add_argument("--no-fleet", action="store_true")
    ap.add_argument("--fleet-hosts", type=int, default=100_000)
    ap.add_argument("--fleet-devices", type=int, default=100, help="mean devices per host")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the CPU baselines (0 = every core this process may run on; where the "
                         "box sets OMP_NUM_THREADS as its CPU share, that many)")
    return ap.parse_args()


def host_cores():
    """(cores this process may run on, threads the CPU baselines use).  On a shared machine the
    affinity mask shows every core of the host while the job's share is what OMP_NUM_THREADS
    says (the GPU boxes set it): more threads than the share only queue up."""
    online = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    use = CPU_THREADS
    if use <= 0:
        try:
            use = int(os.environ.get("OMP_NUM_THREADS", "0"))
        except ValueError:
            use = 0
        use = min(online, use) if use > 0 else online
    return online, max(1, min(use, 256))


CPU_THREADS = 0


def roofline_valu():
    """The tile kernel's own roofline, from the SQ counters on file (profiles/k1_tiles_sq_counters.json,
    10^6 strings x 30 000 patterns): the kernel is bound by instruction issue, not by HBM.
    SQ_ACTIVE_INST_VALU counts in units of four cycles and SQ_BUSY_CYCLES once per shader engine
    (32 of them); 1024 SIMDs."""
    try:
        with open(os.path.join(ROOT, "profiles", "k1_tiles_sq_counters.json")) as f:
            rec = json.load(f)
        c = rec["counters"]["k1_tiles_sq1"]
        busy = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024.0 * c["SQ_BUSY_CYCLES"] / 32.0)
        lanes = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
        return {"bound": "valu issue", "valu_busy_frac": busy, "lanes_active_of_64": lanes,
                "frac_of_lane_throughput": busy * lanes / 64.0,
                "valu_per_string": rec["derived"]["valu_instructions_per_string"],
                "source": "profiles/k1_tiles_sq_counters.json (rocprofv3 --pmc on " + rec["workload"] + "), on file, "
                          "not measured in this run"}
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        return None


def k1_algorithmic_bytes(S, s_off, A, a_off):
    """every input byte once + first_idx and count: |S| + 4(N+1) + |A| + 4(P+1) + 4P + 8N"""
    N, P = len(s_off) - 1, len(a_off) - 1
    return len(S) + 4 * (N + 1) + len(A) + 4 * (P + 1) + 4 * P + 8 * N


def measured_traffic(N, P):
    """HBM bytes per K1 launch from rocprofv3 PMC passes of this same command
    (FETCH_SIZE doubled, as gfx950 reports half of a streamed read, plus
    WRITE_SIZE; both in KiB), kept in profiles/k1_traffic.json by
    scripts/profile_digest.py.  None when no measurement of this workload is on file."""
    path = os.path.join(ROOT, "profiles", "k1_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        if rec.get("strings") == N and rec.get("patterns") == P:
            return rec["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def measured_traffic_k2(algorithmic_bytes):
    """HBM bytes per dependency-flag call (both kernels) from the PMC passes on file in
    profiles/k2_traffic.json; None unless they were taken on this same graph."""
    try:
        with open(os.path.join(ROOT, "profiles", "k2_traffic.json")) as f:
            rec = json.load(f)
        if rec.get("algorithmic_bytes_per_call") == algorithmic_bytes:
            return rec["hbm_bytes_per_call"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline_k1(A, a_off, S, s_off, budget_s):
    """Oracle (glibc fnmatch double loop) on the same workload, all host cores
    this process may use, repeated whole passes until the budget is spent."""
    import oracle

    N, P = len(s_off) - 1, len(a_off) - 1
    online, cores = host_cores()
    oracle.oracle()  # load before timing
    bounds = [N * k // cores for k in range(cores + 1)]

    def work(k):
        if bounds[k + 1] > bounds[k]:
            oracle.alias_match(A, a_off, S, s_off, i0=bounds[k], i1=bounds[k + 1])

    passes, t0 = 0, time.perf_counter()
    while True:
        th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        passes += 1
        el = time.perf_counter() - t0
        if el >= budget_s or passes >= 1000:
            break
    res = {
        "value": passes * N * P / el,
        "unit": "pairs/s",
        "cores": cores,
        "threads_used": cores,
        "cores_online": online,
        "kind": "port",
        "sample": f"{passes} whole passes of the same {N}x{P} workload, glibc fnmatch(p,s,0) double loop, "
                  f"{cores} threads over strings, {el:.1f} s",
    }
    # beside the naive loop: the same index, screen and matchers the GPU uses, on the host cores
    # (the product's pattern compiler and matcher routines built for the CPU by the test twin)
    s_off32 = np.ascontiguousarray(s_off, dtype=np.uint32)
    S8 = np.ascontiguousarray(S if len(S) else np.zeros(1, np.uint8), dtype=np.uint8)
    table = oracle.TwinTable(A, a_off)
    first = np.zeros(N, dtype=np.uint32)
    count = np.zeros(N, dtype=np.uint32)

    def work_indexed(k):
        if bounds[k + 1] > bounds[k]:
            table.match(S8, s_off32, first, count, bounds[k], bounds[k + 1])

    passes, t0 = 0, time.perf_counter()
    while True:
        th = [threading.Thread(target=work_indexed, args=(k,)) for k in range(cores)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        passes += 1
        el = time.perf_counter() - t0
        if el >= budget_s / 4 or passes >= 20000:
            break
    table.close()
    res["indexed"] = {
        "value": passes * N * P / el,
        "unit": "pairs/s",
        "cores": cores,
        "sample": f"{passes} passes of the same workload through the prefix index + gram screen + segment "
                  f"matcher on the CPU (table compiled once, not timed), {el:.1f} s",
    }
    return res


def bracket(step, fence, ctx, steps):
    """EXACTLY `steps` steps between two fences (barrier + torch.cuda.synchronize); returns
    (wall seconds, device milliseconds from HIP events on the context's stream)"""
    fence()
    t0 = time.perf_counter()
    ctx.timer_begin()
    for _ in range(steps):
        step()
    dev_ms = ctx.timer_end()
    fence()
    return time.perf_counter() - t0, dev_ms


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    global CPU_THREADS
    CPU_THREADS = args.cpu_threads
    if args.gpus > 1 and world == 1:
        # Called without a launcher: start the ranks as child processes, one per GPU.  Nothing in
        # this process has touched the GPU yet (torch is not even imported), which is the only
        # time this is allowed; rank 0's line goes to the stdout the children inherit.
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    if os.environ.get("MCX_BENCH_DRYRUN") == "1":
        # launch plumbing only (no GPU, no library): rendezvous over gloo, one line from rank 0
        if world > 1:
            import torch.distributed as dist

            dist.init_process_group("gloo")
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup}))
        return

    import torch

    from montecarlo_amd import Context, synth

    dist = None
    # rehearsal of the multi-rank path on a box with fewer GPUs than ranks: every rank on GPU 0,
    # rendezvous and reductions over gloo on host tensors
    rehearse = os.environ.get("MCX_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    stream = torch.cuda.Stream(device=local_rank)
    ctx = Context(local_rank, stream=stream.cuda_stream)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        out = bench_sharded(ctx, synth, args, torch, dist, stream, fence, rank, world, local_rank, rehearse)
    else:
        out = bench_one_gpu(ctx, synth, args, torch, stream, fence, local_rank)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


METRIC = "device x alias glob matches/sec + dep-closure edges/sec, 1 GPU vs host C"


def bench_one_gpu(ctx, synth, args, torch, stream, fence, local_rank):
    """configs[1]: the headline line, and beside it the blocks for configs[2], [3] and [4]"""
    N, P = args.strings, args.patterns
    A, a_off, a_mod = synth.alias_table(args.seed, P)
    S, s_off = synth.modalias_strings(args.seed, P, N)
    ctx.alias_table_load(A, a_off, a_mod)   # the first load also pays for the code object
    t0 = time.perf_counter()
    ctx.alias_table_load(A, a_off, a_mod)
    table_load_ms = (time.perf_counter() - t0) * 1e3
    d_S, d_off = ctx.to_device(S), ctx.to_device(s_off)
    t_first = torch.empty(N, dtype=torch.int32, device=f"cuda:{local_rank}")
    t_count = torch.empty(N, dtype=torch.int32, device=f"cuda:{local_rank}")

    def step():
        ctx.alias_match_dev(d_S.ptr, d_off.ptr, N, t_first.data_ptr(), t_count.data_ptr())

    for _ in range(args.warmup):
        step()
    # A bracket of K steps of 10 us each is a fraction of a millisecond: one alone is at the mercy
    # of a single host hiccup.  The bracket is repeated and the median one is reported.
    runs = sorted(bracket(step, fence, ctx, args.steps) for _ in range(max(1, args.repeats)))
    wall, dev_ms = runs[len(runs) // 2]
    from montecarlo_amd import shard

    first, count = shard.as_u32(t_first), shard.as_u32(t_count)
    alg_bytes = k1_algorithmic_bytes(S, s_off, A, a_off)
    dev_s_per_launch = dev_ms / 1e3 / args.steps
    achieved = alg_bytes / dev_s_per_launch / 1e9
    out = {
        "metric": METRIC,
        "value": N * P * args.steps / wall,
        "unit": "pairs/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": f"configs[1]: {N} modalias strings x {P} alias patterns, seed {args.seed}, "
                        "strings and table resident in HBM",
            "strings_per_gpu": N,
            "patterns": P,
            "sharding": "none",
        },
        "timing": {
            "brackets": len(runs),
            "reported": "median bracket of `steps` steps (each between barrier + synchronize)",
            "ms_per_step_min": runs[0][0] * 1e3 / args.steps,
            "ms_per_step_max": runs[-1][0] * 1e3 / args.steps,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": measured_traffic(N, P),
            "kernel": "k1_match_small, one launch per step, device time from HIP events on its stream",
            "algorithmic_bytes_per_launch": alg_bytes,
            "device_us_per_launch": dev_s_per_launch * 1e6,
        },
        "matched_strings": int((count > 0).sum()),
        "table_load_ms": table_load_ms,   # host compile of the table + upload, once per table
    }
    if not args.no_cpu:
        import oracle

        n_chk = min(N, 64)
        o_first, o_count = oracle.alias_match(A, a_off, S, s_off, i0=0, i1=n_chk)
        if not (np.array_equal(first[:n_chk], o_first) and np.array_equal(count[:n_chk], o_count)):
            raise SystemExit("bench: GPU result differs from the oracle")
        out["cpu_baseline"] = cpu_baseline_k1(A, a_off, S, s_off, args.cpu_seconds)
    for b in (d_S, d_off):
        b.free()
    out["lists"] = bench_lists(ctx, synth, args, with_cpu=not args.no_cpu)
    if not args.no_k2:
        out["k2"] = bench_k2(ctx, synth, args, with_cpu=not args.no_cpu)
    if not args.no_fleet:
        out["fleet"] = bench_fleet(ctx, synth, args, with_cpu=not args.no_cpu)
    if not args.no_configs3:
        out["configs3"] = bench_configs3(ctx, synth, args, with_cpu=not args.no_cpu)
    return out


def bench_lists(ctx, synth, args, with_cpu=False):
    """Match lists (every matching table position per string, kmod's order) against first / count
    alone, 70 000 strings on the table that is loaded (configs[1]'s)."""
    N, P = 70000, args.patterns
    S, s_off = synth.modalias_strings(args.seed, P, N)
    d_S, d_off = ctx.to_device(S), ctx.to_device(s_off)
    d_f, d_c, d_l = ctx.alloc(4 * N), ctx.alloc(4 * N), ctx.alloc(4 * (N + 1))
    total = ctx.alias_match_list_dev(d_S.ptr, d_off.ptr, N, 0, d_l.ptr, None, 0, size_only=True)
    d_i = ctx.alloc(4 * max(total, 1))
    res = {"workload": f"{N} modalias strings x {P} alias patterns, seed {args.seed}, resident in HBM",
           "entries": int(total)}
    for name, fn in (("first_count_us", lambda: ctx.alias_match_dev(d_S.ptr, d_off.ptr, N, d_f.ptr, d_c.ptr)),
                     ("lists_us", lambda: ctx.alias_match_list_dev(d_S.ptr, d_off.ptr, N, 0, d_l.ptr, d_i.ptr, total)),
                     ("lists_first3_us",
                      lambda: ctx.alias_match_list_dev(d_S.ptr, d_off.ptr, N, 3, d_l.ptr, d_i.ptr, total))):
        for _ in range(3):
            fn()
        ctx.sync()
        ctx.timer_begin()
        for _ in range(20):
            fn()
        res[name] = ctx.timer_end() / 20 * 1e3
    res["lists_over_first_count"] = res["lists_us"] / res["first_count_us"]
    if with_cpu:
        import oracle

        ctx.alias_match_list_dev(d_S.ptr, d_off.ptr, N, 0, d_l.ptr, d_i.ptr, total)
        off, idx = d_l.download(np.uint32, N + 1), d_i.download(np.uint32, total)
        n_chk = 64
        A, a_off, _ = synth.alias_table(args.seed, P)
        o_off, o_idx = oracle.alias_match_list(A, a_off, S[: s_off[n_chk]], s_off[: n_chk + 1])
        if not (np.array_equal(off[: n_chk + 1], o_off) and np.array_equal(idx[: o_off[n_chk]], o_idx)):
            raise SystemExit("bench: match lists differ from the oracle")
    for b in (d_S, d_off, d_f, d_c, d_l, d_i):
        b.free()
    return res


def configs3_inputs(synth, args, j0=0, j1=None):
    P, N, seed = args.c3_patterns, args.c3_strings, 4
    A, a_off, a_mod = synth.alias_table(seed, P, j0, P if j1 is None else j1)
    S, s_off = synth.modalias_strings(seed, P, N)
    return A, a_off, a_mod, S, s_off


def bench_configs3(ctx, synth, args, with_cpu=False):
    """configs[3] on ONE GPU: the whole 3M-pattern table against the whole 1M-string batch (it
    fits: half a gigabyte).  The CPU side is a sample, extrapolated and labelled so."""
    P, N = args.c3_patterns, args.c3_strings
    t0 = time.perf_counter()
    A, a_off, a_mod, S, s_off = configs3_inputs(synth, args)
    gen_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    ctx.alias_table_load(A, a_off, a_mod)
    load_s = time.perf_counter() - t0
    d_S, d_off = ctx.to_device(S), ctx.to_device(s_off)
    d_f, d_c = ctx.alloc(4 * N), ctx.alloc(4 * N)
    for _ in range(2):
        ctx.alias_match_dev(d_S.ptr, d_off.ptr, N, d_f.ptr, d_c.ptr)
    ctx.sync()
    reps = 5
    ctx.timer_begin()
    for _ in range(reps):
        ctx.alias_match_dev(d_S.ptr, d_off.ptr, N, d_f.ptr, d_c.ptr)
    ms = ctx.timer_end() / reps
    first, count = d_f.download(np.uint32, N), d_c.download(np.uint32, N)
    alg = k1_algorithmic_bytes(S, s_off, A, a_off)
    res = {
        "workload": f"configs[3] on one GPU: {N} modalias strings x {P} alias patterns, seed 4, resident in HBM",
        "ms_per_pass": ms,
        "pairs_per_s": N * P / (ms / 1e3),
        "algorithmic_bytes": alg,
        "achieved_GBs": alg / (ms / 1e3) / 1e9,
        "frac_of_hbm_peak": alg / (ms / 1e3) / 1e9 / HBM_PEAK_GBS,
        "matched_strings": int((count > 0).sum()),
        "host_seconds": {"generate_inputs": gen_s, "compile_and_load_table": load_s},
        "roofline_valu": roofline_valu(),
    }
    if with_cpu:
        import oracle

        # glibc loop: about 0.1 s per string against 3M patterns; a sample on all cores, extrapolated
        online, cores = host_cores()
        per = max(1, int(args.c3_cpu_strings) // cores)
        n_s = per * cores
        got = [None] * cores

        def work(k):
            got[k] = oracle.alias_match(A, a_off, S, s_off, i0=k * per, i1=(k + 1) * per)

        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        o_first = np.concatenate([g[0] for g in got])
        o_count = np.concatenate([g[1] for g in got])
        if not (np.array_equal(first[:n_s], o_first) and np.array_equal(count[:n_s], o_count)):
            raise SystemExit("bench: configs[3] result differs from the oracle")
        res["cpu_baseline"] = {
            "value": n_s * P / el, "unit": "pairs/s", "cores": cores, "threads_used": cores, "cores_online": online,
            "kind": "port",
            "sample": f"the first {n_s} of the {N} strings through the glibc fnmatch loop against all {P} "
                      f"patterns, {cores} threads, {el:.1f} s; EXTRAPOLATED to the whole batch: "
                      f"{el * N / n_s / 3600:.1f} hours",
            "extrapolated_seconds_per_pass": el * N / n_s,
        }
        # the fair CPU number: the same index, screen and matchers on the host cores (the twin), on
        # 1 % of the strings, table compiled once and not timed
        n_i = max(cores, N // 100)
        t0 = time.perf_counter()
        table = oracle.TwinTable(A, a_off)
        compile_s = time.perf_counter() - t0
        i_first = np.zeros(N, dtype=np.uint32)
        i_count = np.zeros(N, dtype=np.uint32)
        S8, s32 = np.ascontiguousarray(S, dtype=np.uint8), np.ascontiguousarray(s_off, dtype=np.uint32)
        cut = [n_i * k // cores for k in range(cores + 1)]
        t0 = time.perf_counter()
        th = [threading.Thread(target=table.match, args=(S8, s32, i_first, i_count, cut[k], cut[k + 1]))
              for k in range(cores) if cut[k + 1] > cut[k]]
        for t in th:
            t.start()
        for t in th:
            t.join()
        el_i = time.perf_counter() - t0
        table.close()
        if not (np.array_equal(first[:n_i], i_first[:n_i]) and np.array_equal(count[:n_i], i_count[:n_i])):
            raise SystemExit("bench: configs[3] result differs from the indexed CPU twin")
        res["cpu_baseline"]["indexed"] = {
            "value": n_i * P / el_i, "unit": "pairs/s", "cores": cores,
            "sample": f"the first {n_i} of the {N} strings (1 %) through the prefix index + gram screen + segment "
                      f"matcher on the CPU, {cores} threads, {el_i:.2f} s (table compiled once in {compile_s:.1f} s, "
                      f"not timed); EXTRAPOLATED to the whole batch: {el_i * N / n_i:.1f} s",
            "extrapolated_seconds_per_pass": el_i * N / n_i,
        }
    for b in (d_S, d_off, d_f, d_c):
        b.free()
    return res


def bench_sharded(ctx, synth, args, torch, dist, stream, fence, rank, world, local_rank, rehearse):
    """--gpus N > 1: configs[3], 1M strings x 3M patterns.  Headline: the alias table cut into N
    contiguous slices (one per GPU), every GPU matching all strings against its slice, one MIN and
    one SUM all-reduce of 1M words per step over RCCL.  Beside it in the same line: the collective
    alone, the kernel alone, the string-sharded layout (whole table everywhere, no collective) and,
    on rank 0, the whole problem on one GPU."""
    from montecarlo_amd import shard

    P, N = args.c3_patterns, args.c3_strings
    dev = f"cuda:{local_rank}"
    j0, j1 = shard.shard_range(P, rank, world)
    A, a_off, a_mod, S, s_off = configs3_inputs(synth, args, j0, j1)
    ctx.alias_table_load(A, a_off, a_mod)
    d_S, d_off = ctx.to_device(S), ctx.to_device(s_off)
    t_first = torch.empty(N, dtype=torch.int32, device=dev)
    t_count = torch.empty(N, dtype=torch.int32, device=dev)

    def combine():
        if rehearse:  # gloo reduces host tensors: the rehearsal takes the results through the host
            stream.synchronize()
            f, c = t_first.cpu(), t_count.cpu()
            shard.combine_pattern_shards(f, c, j0)
            t_first.copy_(f)
            t_count.copy_(c)
        else:
            with torch.cuda.stream(stream):
                shard.combine_pattern_shards(t_first, t_count, j0)

    def match():
        ctx.alias_match_dev(d_S.ptr, d_off.ptr, N, t_first.data_ptr(), t_count.data_ptr())

    def step():
        match()
        combine()

    def maxed(*vals):
        t = torch.tensor(vals, dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    for _ in range(args.warmup):
        step()
    wall, dev_ms = maxed(*bracket(step, fence, ctx, args.steps))
    first, count = shard.as_u32(t_first).copy(), shard.as_u32(t_count).copy()
    k_wall, k_dev_ms = maxed(*bracket(match, fence, ctx, args.steps))   # the kernel alone
    c_wall, c_dev_ms = maxed(*bracket(combine, fence, ctx, args.steps))  # the two all-reduces alone
    alg_bytes = k1_algorithmic_bytes(S, s_off, A, a_off)
    k_s = k_dev_ms / 1e3 / args.steps
    out = {
        "metric": METRIC,
        "value": N * P * args.steps / wall,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": f"configs[3]: {N} modalias strings x {P} alias patterns, seed 4, alias table cut into "
                        f"{world} contiguous slices (one per GPU), MIN + SUM all-reduce of {N} words per step",
            "strings": N,
            "patterns": P,
            "sharding": "patterns",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": alg_bytes / k_s / 1e9,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": alg_bytes / k_s / 1e9 / HBM_PEAK_GBS,
            "traffic": None,
            "kernel": "k1_match_tiles on this rank's table slice, device time from HIP events, max over ranks",
            "algorithmic_bytes_per_launch": alg_bytes,
            "device_us_per_launch": k_s * 1e6,
        },
        "pattern_sharded": {
            "ms_per_step": wall * 1e3 / args.steps,
            "kernel_alone_ms": k_wall * 1e3 / args.steps,
            "allreduce_alone_ms": c_wall * 1e3 / args.steps,
            "allreduce_bytes_per_step": 8 * N,
        },
        "matched_strings": int((count > 0).sum()),
    }
    # the other layout: whole table on every GPU, every GPU its own N / world strings, no collective
    for b in (d_S, d_off):
        b.free()
    A, a_off, a_mod, S_all, s_all = configs3_inputs(synth, args)
    ctx.alias_table_load(A, a_off, a_mod)
    i0, i1 = shard.shard_range(N, rank, world)
    S_r, s_r = synth.modalias_strings(4, P, N, i0=i0, i1=i1)
    d_S, d_off = ctx.to_device(S_r), ctx.to_device(s_r)
    n_r = i1 - i0

    def match_own():
        ctx.alias_match_dev(d_S.ptr, d_off.ptr, n_r, t_first.data_ptr(), t_count.data_ptr())

    for _ in range(args.warmup):
        match_own()
    s_wall, _ = maxed(*bracket(match_own, fence, ctx, args.steps))
    own_first = shard.as_u32(t_first)[:n_r].copy()
    out["string_sharded"] = {
        "value": N * P * args.steps / s_wall, "unit": "pairs/s", "ms_per_step": s_wall * 1e3 / args.steps,
        "note": "whole table on every GPU, N / n_gpus strings each, no data-path collective",
        "same_results_as_pattern_sharded": bool(np.array_equal(own_first, first[i0:i1])),
    }
    for b in (d_S, d_off):
        b.free()
    if rank == 0:
        # one GPU, whole problem: the reference the sharded figures are speed-ups of
        d_S, d_off = ctx.to_device(S_all), ctx.to_device(s_all)

        def match_all():
            ctx.alias_match_dev(d_S.ptr, d_off.ptr, N, t_first.data_ptr(), t_count.data_ptr())

        match_all()
        ctx.sync()
        ctx.timer_begin()
        for _ in range(3):
            match_all()
        one_ms = ctx.timer_end() / 3
        same = np.array_equal(shard.as_u32(t_first), first) and np.array_equal(shard.as_u32(t_count), count)
        out["one_gpu"] = {"ms_per_pass": one_ms, "value": N * P / (one_ms / 1e3), "unit": "pairs/s",
                          "same_results_as_pattern_sharded": bool(same)}
        out["pattern_sharded"]["speedup_vs_one_gpu"] = one_ms / (wall * 1e3 / args.steps)
        out["string_sharded"]["speedup_vs_one_gpu"] = one_ms / (s_wall * 1e3 / args.steps)
        for b in (d_S, d_off):
            b.free()
    return out


def bench_k2(ctx, synth, args, with_cpu=False):
    """configs[2]: flags and closure over the synthetic dependency DAG; with_cpu adds the
    oracle's single-threaded C passes over the same graph (the host-C side of the metric)."""
    V = args.graph_vertices
    rowptr, col = synth.dep_graph(3, V, args.graph_degree)
    E = len(col)
    flags = synth.dep_flags(3, rowptr, col)
    seeds, nseed = synth.seed_bitmap(3, flags, 16)
    ctx.dep_graph_load(rowptr, col)
    W = (V + 31) // 32
    d_in, d_seed = ctx.to_device(flags), ctx.to_device(seeds)
    d_out, d_ref, d_clo = ctx.alloc(V), ctx.alloc(4 * V), ctx.alloc(4 * W)
    reps = 20
    for _ in range(10):
        ctx.dep_flags_dev(d_in.ptr, d_out.ptr, d_ref.ptr)
    ctx.sync()
    brackets = []
    for _ in range(9):   # median of nine brackets of 50 calls, device time from HIP events
        ctx.timer_begin()
        for _ in range(50):
            ctx.dep_flags_dev(d_in.ptr, d_out.ptr, d_ref.ptr)
        brackets.append(ctx.timer_end() / 50)
    flags_ms = sorted(brackets)[len(brackets) // 2]
    alg_flags = 4 * (V + 1) + 4 * E + V + V
    res = {
        "workload": f"configs[2]: V={V} modules, E={E} direct-dependency edges, seed 3, "
                    f"{int((flags & 1).sum())} loaded",
        "dep_flags": {
            "edges_per_s": E / (flags_ms / 1e3),
            "ms": flags_ms,
            "algorithmic_bytes": alg_flags,
            "achieved_GBs": alg_flags / (flags_ms / 1e3) / 1e9,
            "frac_of_hbm_peak": alg_flags / (flags_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
            "traffic": measured_traffic_k2(alg_flags),
            "note": "algorithmic bytes count 4 bytes per holder id; the kernel streams a second copy of the "
                    "ids kept at 20 bits each (V <= 2^20), so its HBM traffic is below them",
        },
    }
    moved = res["dep_flags"]["traffic"]
    if moved:
        # what the memory system really carried (PMC traffic on file) over the measured time
        res["dep_flags"]["moved_GBs"] = moved / (flags_ms / 1e3) / 1e9
        res["dep_flags"]["frac_moved"] = moved / (flags_ms / 1e3) / 1e9 / HBM_PEAK_GBS
    # closure: 16 seeds (what an operator asks for one device's drivers: latency bound) and the
    # large-frontier case, seeds = every BOUND module (in the deps direction that is the set of
    # modules that cannot be unloaded safely)
    bound_seeds = np.zeros(W, dtype=np.uint32)
    ids = np.nonzero((flags & 3) == 3)[0]
    np.bitwise_or.at(bound_seeds, ids >> 5, np.uint32(1) << (ids & 31).astype(np.uint32))
    d_bound = ctx.to_device(bound_seeds)
    cases = (("closure_holders", 0, d_seed, nseed), ("closure_deps", 1, d_seed, nseed),
             ("closure_holders_of_bound", 0, d_bound, len(ids)), ("closure_deps_of_bound", 1, d_bound, len(ids)))
    for name, direction, d_s, ns in cases:
        # three calls first: a search places its launches by the sizes the last search in that
        # direction had; then the median of five runs of ten calls (a call returns when the search
        # is over, the result bitmap follows in stream order: one synchronisation per run)
        for _ in range(3):
            ctx.dep_closure_dev(d_in.ptr, d_s.ptr, d_clo.ptr, direction)
        ctx.sync()
        runs = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(10):
                edges, levels = ctx.dep_closure_dev(d_in.ptr, d_s.ptr, d_clo.ptr, direction)
            ctx.sync()
            runs.append((time.perf_counter() - t0) * 1e3 / 10)
        ms = sorted(runs)[2]
        # B2b = row pointers + scanned adjacency entries + the bitmap once per level and side
        alg = 4 * (V + 1) + 4 * edges + (V // 8) * (2 * levels + 2)
        gbs = alg / (ms / 1e3) / 1e9 if ms > 0 else 0.0
        res[name] = {
            "seeds": int(ns),
            "edges_scanned": edges,
            "levels": levels,
            "ms": ms,
            "edges_per_s": edges / (ms / 1e3) if ms > 0 else 0.0,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes": int(alg), "traffic": None,
                         "note": "wall time of the whole call (launches, level look-ups) over algorithmic bytes"},
        }
    d_bound.free()
    for b in (d_in, d_seed, d_out, d_ref, d_clo):
        b.free()
    if with_cpu:
        import oracle

        t0 = time.perf_counter()
        n = 0
        while n < 3 or time.perf_counter() - t0 < 1.0:
            oracle.dep_flags(rowptr, col, flags)
            n += 1
        cpu_flags_ms = (time.perf_counter() - t0) * 1e3 / n
        cpu = {"cores": 1, "kind": "port",
               "sample": f"{n} whole passes of the same graph, edge scan + queue BFS in C, one thread",
               "dep_flags": {"ms": cpu_flags_ms, "edges_per_s": E / (cpu_flags_ms / 1e3)}}
        # holders of every module, built once and not timed (the GPU side builds its own at graph
        # load): the holders-direction search is then a forward search over this graph
        rrow, rcol = oracle.holders_graph(rowptr, col)
        # the same flags from several threads, each counting the loaded holders of its own modules
        from concurrent.futures import ThreadPoolExecutor

        online, threads = host_cores()
        cut = oracle.holders_cut(rrow, threads)
        with ThreadPoolExecutor(max_workers=threads) as pool:
            oracle.dep_flags_threads(rrow, rcol, flags, threads, pool, cut)
            t0 = time.perf_counter()
            n = 0
            while n < 3 or time.perf_counter() - t0 < 1.0:
                oracle.dep_flags_threads(rrow, rcol, flags, threads, pool, cut)
                n += 1
            ms = (time.perf_counter() - t0) * 1e3 / n
        cpu["dep_flags_threads"] = {"cores": threads, "cores_online": online, "ms": ms, "edges_per_s": E / (ms / 1e3),
                                    "note": "over the graph of holders, built once and not timed"}
        for name, graph in (("closure_holders", (rrow, rcol)), ("closure_deps", (rowptr, col))):
            t0 = time.perf_counter()
            n = 0
            while n < 3 or time.perf_counter() - t0 < 0.5:
                _, o_edges, _ = oracle.dep_closure(graph[0], graph[1], flags, seeds, 1)
                n += 1
            ms = (time.perf_counter() - t0) * 1e3 / n
            if o_edges != res[name]["edges_scanned"]:
                raise SystemExit("bench: closure differs from the oracle")
            cpu[name] = {"ms": ms, "edges_per_s": o_edges / (ms / 1e3) if ms > 0 else 0.0}
        res["cpu_baseline"] = cpu
    return res


def bench_fleet(ctx, synth, args, with_cpu=False):
    """configs[4]: alias match -> used drivers -> loaded set -> holders -> root filter for a
    whole fleet, inputs resident in HBM; with_cpu times the oracle on the first hosts of the
    same fleet (alias part by the indexed CPU twin: the glibc loop would take minutes)."""
    from montecarlo_amd import safe_module_names

    H, P, seed = args.fleet_hosts, args.patterns, 5
    A, a_off, a_mod = synth.alias_table(seed, P)
    V = synth.alias_modules(P)
    rowptr, col = synth.dep_graph(seed + 1, V, args.graph_degree)
    dev_cnt, extra_off, extra_ids = synth.fleet_hosts(seed + 2, V, H, args.fleet_devices, 8)
    dev_off = np.zeros(H + 1, dtype=np.uint32)
    dev_off[1:] = np.cumsum(dev_cnt)
    N = int(dev_off[H])
    S, s_off = synth.modalias_strings(seed + 3, P, N)
    safe = safe_module_names(synth.module_names(seed + 4, V))
    ctx.alias_table_load(A, a_off, a_mod)
    ctx.dep_graph_load(rowptr, col)
    bufs = [ctx.to_device(x) for x in (S, s_off, dev_off, extra_off, extra_ids, safe)]
    d_S, d_soff, d_dev, d_xoff, d_xid, d_safe = bufs
    d_roff = ctx.alloc(4 * (H + 1))
    X = len(extra_ids)
    total = ctx.fleet_roots_dev(d_S.ptr, d_soff.ptr, N, d_dev.ptr, H, d_xoff.ptr, d_xid.ptr, X, d_safe.ptr,
                                d_roff.ptr, None, 0, size_only=True)
    d_rid = ctx.alloc(4 * max(total, 1))
    reps = 3
    for _ in range(reps + 1):  # first pass untimed
        if _ == 1:
            ctx.sync()
            t0 = time.perf_counter()
        ctx.fleet_roots_dev(d_S.ptr, d_soff.ptr, N, d_dev.ptr, H, d_xoff.ptr, d_xid.ptr, X, d_safe.ptr,
                            d_roff.ptr, d_rid.ptr, total)
    ctx.sync()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    got_ids = d_rid.download(np.uint32, total)

    def root_ids_head(n):
        return got_ids[:n]

    d_f, d_c = ctx.alloc(4 * N), ctx.alloc(4 * N)
    ctx.alias_match_dev(d_S.ptr, d_soff.ptr, N, d_f.ptr, d_c.ptr)
    ctx.timer_begin()
    ctx.alias_match_dev(d_S.ptr, d_soff.ptr, N, d_f.ptr, d_c.ptr)
    k1_ms = ctx.timer_end()
    for b in bufs[3:] + [d_roff, d_rid, d_f, d_c]:
        b.free()
    # the step before the match: which devices a host's list would show (per-host limit 64,
    # the value the reference's UI passes), and the packing of their strings
    rec = synth.device_records(seed + 5, S, s_off)
    arena, off = synth.driver_names_packed()
    d_rec, d_arena, d_off = ctx.to_device(rec), ctx.to_device(arena), ctx.to_device(off)
    d_v, d_kept = ctx.alloc(N), ctx.alloc(4 * (H + 1))
    D = len(off) - 1
    K = ctx.device_filter_dev(d_rec.ptr, N, d_arena.ptr, d_off.ptr, D, d_dev.ptr, H, 64, d_v.ptr, d_kept.ptr)
    d_S2, d_soff2 = ctx.alloc(len(S)), ctx.alloc(4 * (K + 1))
    kept_bytes = ctx.fleet_compact_dev(d_v.ptr, d_S.ptr, d_soff.ptr, N, d_dev.ptr, H, d_kept.ptr, K, d_S2.ptr,
                                       d_soff2.ptr)
    ctx.timer_begin()
    for _ in range(reps):
        ctx.device_filter_dev(d_rec.ptr, N, d_arena.ptr, d_off.ptr, D, d_dev.ptr, H, 64, d_v.ptr, d_kept.ptr)
    filter_ms = ctx.timer_end() / reps
    ctx.timer_begin()
    for _ in range(reps):
        ctx.fleet_compact_dev(d_v.ptr, d_S.ptr, d_soff.ptr, N, d_dev.ptr, H, d_kept.ptr, K, d_S2.ptr, d_soff2.ptr,
                              None, want_bytes=False)
    compact_ms = ctx.timer_end() / reps
    cpu = None
    if with_cpu:
        import oracle

        # hosts are independent: the per-host loop on every core, a range of hosts each
        online, cores = host_cores()
        Hs = min(H, 500 * cores)
        Ns = int(dev_off[Hs])
        hw = ctx.alias_hw_flags(V)
        table = oracle.TwinTable(A, a_off)   # the alias part by the indexed twin, compiled once, not timed
        S8, s32 = np.ascontiguousarray(S, dtype=np.uint8), np.ascontiguousarray(s_off, dtype=np.uint32)
        f_s, c_s = np.zeros(N, dtype=np.uint32), np.zeros(N, dtype=np.uint32)
        hcut = [Hs * k // cores for k in range(cores + 1)]
        parts = [None] * cores

        def hosts(k):
            h0, h1 = hcut[k], hcut[k + 1]
            if h1 <= h0:
                return
            i0, i1 = int(dev_off[h0]), int(dev_off[h1])
            table.match(S8, s32, f_s, c_s, i0, i1)
            f = f_s[i0:i1]
            drv = np.where(f != 0xFFFFFFFF, a_mod[np.minimum(f, len(a_mod) - 1)], 0xFFFFFFFF).astype(np.uint32)
            x0, x1 = int(extra_off[h0]), int(extra_off[h1])
            parts[k] = oracle.fleet_roots(rowptr, col, drv, dev_off[h0: h1 + 1] - dev_off[h0],
                                          extra_off[h0: h1 + 1] - extra_off[h0], extra_ids[x0:x1], safe, hw)

        t0 = time.perf_counter()
        th = [threading.Thread(target=hosts, args=(k,)) for k in range(cores)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        fleet_s = time.perf_counter() - t0
        table.close()
        o_ids = np.concatenate([p[1] for p in parts if p is not None])
        if not np.array_equal(o_ids, root_ids_head(o_ids.size)):
            raise SystemExit("bench: fleet lists differ from the oracle")
        t0 = time.perf_counter()
        Nr = min(N, 1_000_000)
        Hr = int(np.searchsorted(dev_off, Nr, side="right")) - 1
        oracle.device_filter(rec[: int(dev_off[Hr])], arena, off, dev_off[: Hr + 1], 64)
        filt_s = time.perf_counter() - t0
        cpu = {"cores": cores, "threads_used": cores, "cores_online": online, "kind": "port",
               "sample": f"first {Hs} hosts ({Ns} devices) through the per-host loop, {cores} threads (a range of "
                         f"hosts each), alias part by the indexed twin; device list predicates over the first "
                         f"{int(dev_off[Hr])} records on one thread",
               "fleet_hosts_per_s": Hs / fleet_s, "fleet_devices_per_s": Ns / fleet_s,
               "device_filter_records_per_s": int(dev_off[Hr]) / filt_s}
    filter_bytes = 16 * N + 2 * N + 8 * (H + 1)
    compact_bytes = N + 4 * (N + 1) + 8 * (H + 1) + 2 * kept_bytes + 4 * (K + 1)
    for b in bufs[:3] + [d_rec, d_arena, d_off, d_v, d_kept, d_S2, d_soff2]:
        b.free()
    return {
        **({"cpu_baseline": cpu} if cpu else {}),
        "device_filter": {
            "listed_devices": K,
            "filter_ms": filter_ms,
            "filter_GBps": filter_bytes / (filter_ms / 1e3) / 1e9,
            "compact_ms": compact_ms,
            "compact_GBps": compact_bytes / (compact_ms / 1e3) / 1e9,
            "bytes": {"filter": int(filter_bytes), "compact": int(compact_bytes)},
        },
        "workload": f"configs[4]: {H} hosts, {N} devices, {P} alias patterns, {V} modules, {len(col)} edges, "
                    f"{X} extra loaded modules, seed {seed}",
        "ms_per_pass": ms,
        "devices_per_s": N / (ms / 1e3),
        "hosts_per_s": H / (ms / 1e3),
        "listed_modules": total,
        "alias_match_ms": k1_ms,
        "roofline_valu": roofline_valu(),
        "pairs_per_s": N * P / (k1_ms / 1e3),
        "input_bytes": int(len(S) + 4 * (N + 1) + 4 * (H + 1) * 2 + 4 * X + V),
    }


if __name__ == "__main__":
    main()
