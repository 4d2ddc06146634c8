#!/usr/bin/env python3
"""Benchmark of the stereo-pair -> disparity-map path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one synthetic 1242x375 stereo pair at D=192 through the whole path: guidance statistics, the
fused single-kernel aggregation (cost build -> integral -> box -> a,b -> integral -> box -> q per strip,
a and b never leave the CU), running WTA of both views, (N > 1: one RCCL MIN all-reduce of the packed
int64 keys, the disparity slices being sharded across ranks), decode, LR check, filling.  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

roofline: the dominant kernel -- the fused aggregation walker (k_v5_walk / k_v4_walk: left + right volume in one
launch) -- against HBM: algorithmic bytes are 8 B per (pixel, disparity) cell of a volume (read raw cost 4 B +
write/consume aggregated cost 4 B) x the cells one launch processes (2 volumes), divided by the
kernel's average launch duration, measured live with HIP events recorded on the launch stream at the stage
boundaries inside the C-ABI (smx_set_timing(2) / smx_stage_times).  The events run in a second pass over the same K
steps: every event record costs a bubble on the queue, and 11 of them per 0.9 ms step would take ~9 % off `value`;
the timed region of the contract (value, ms_per_step) has no event in it.  `operator` repeats the figure for
the whole smx_dev_aggregate_wta_pair call (key presets, guidance statistics, walker, WTA pass), which is what the
lines of rounds 1-3 reported as `frac`.
cpu_baseline: the CPU oracle (port of the reference kernels, 1 thread) timed on this box's host
cores on the same pair -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
ALGO_BYTES_PER_CELL = 8.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", default="exact", choices=["exact", "fast"],
                    help="fast = the non-bit-exact aggregation with wave-parallel row scans (smx_set_agg_path(4)): an "
                         "upper-bound point that is reported separately, never the headline")
    ap.add_argument("--workload", default="kitti", choices=["tsukuba", "kitti", "motorcycle", "4k"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preheat-s", type=float, default=0.3, help="seconds of untimed steps in front of the warmup steps")
    ap.add_argument("--slices-in-flight", type=int, default=None,
                    help="upper bound on the slices of one walker launch (default: all local slices in one launch)")
    ap.add_argument("--source", default="images", choices=["images", "cost"],
                    help="images (the headline): the matching costs are built inside the aggregation kernel from the two images; "
                         "cost: the reference's own data flow (main.cu:80-82 then :133-134, guidedFilter.cu:198-233) -- both raw "
                         "cost volumes materialised in HBM before the timed region, every step reads p and writes / consumes q: "
                         "the literal 8 B per cell of the roofline accounting.  Reported separately, never the headline")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N > 1 exchange step: nccl = RCCL over xGMI, one GPU per rank (what the "
                         "scaling run uses); gloo = the SAME control flow with the keys staged through host memory, ranks may "
                         "share a GPU (rehearsal of the N > 1 path on a one-GPU box; not a performance number)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import stereo_matching_cuda_amd as smx
    from stereo_matching_cuda_amd import synth
    from stereo_matching_cuda_amd.sharded import ShardedPair, shard_range

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"--backend nccl needs one GPU per rank ({world} ranks, {ndev} devices); --backend gloo lets ranks share a GPU")
    dev_index = local_rank % ndev          # (gloo rehearsal: ranks may share a device)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    smx.lib()  # fail loudly if the HIP extension is missing
    if args.mode == "fast":
        smx.check(smx.lib().smx_set_agg_path(4))
    w, h, D = synth.SHAPES[args.workload]
    seed = synth.SEEDS.get(args.workload, 1)
    Il, Ir = synth.gen_pair(w, h, D, seed)
    dl = torch.from_numpy(Il).to(device)
    dr = torch.from_numpy(Ir).to(device)

    sp = ShardedPair(w, h, D, rank=rank, world=world, device=device,
                     slices_in_flight=args.slices_in_flight)
    pipe = sp.pipe
    local_slices = pipe.s_end - pipe.s_begin

    cost_l = cost_r = None
    if args.source == "cost":
        cost_l, cost_r = pipe.cost_volumes(dl, dr)      # resident in HBM like the images; built once, outside every timed region
        torch.cuda.synchronize(device)

    def step(events=None):
        if events is not None:
            es = torch.cuda.Event(enable_timing=True)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            ee = torch.cuda.Event(enable_timing=True)
            es.record()
        if events is not None:
            e0.record()
        # both views per kernel launch; the call presets the WTA keys itself (smx_set_keys_fresh: no smx_dev_init_keys launch)
        pipe.aggregate(dl, dr, cost_l, cost_r)
        if events is not None:
            e1.record()
        if world > 1:
            from stereo_matching_cuda_amd.sharded import allreduce_min_keys_
            allreduce_min_keys_(pipe.keys)
        pipe.finish()
        if events is not None:
            ee.record()
            events.append((es, e0, e1, ee))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # clocks and caches to steady state before anything is counted: ~0.3 s of steps (a 20-step run is 20 ms long, shorter
    # than the power controller's ramp), then the W warmup steps of the contract
    preheat = 0
    tp = time.perf_counter()
    if world > 1:
        # every rank must run the SAME number of steps (each step has a collective in it): a fixed count, not a clock
        for _ in range(200 if args.preheat_s > 0 else 0):
            step()
        torch.cuda.synchronize(device)
        preheat = 200 if args.preheat_s > 0 else 0
    else:
        while time.perf_counter() - tp < args.preheat_s:
            for _ in range(10):
                step()
            torch.cuda.synchronize(device)
            preheat += 10
    for _ in range(args.warmup):
        step()
    fence()
    pipe.check_status()       # a timed-out hand-off would invalidate everything that follows
    import ctypes as C
    from stereo_matching_cuda_amd import _lib

    # ---- the timed region of the contract: exactly K steps between two fences, nothing but the hot path in it (an event
    # record costs a bubble of 5-20 us on this queue: per-step and per-stage events would take ~9 % off a 0.9 ms step)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    pipe.check_status()       # ... and the status word is per call: read it before the next call clears it
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    chunk, walker_launches = pipe.last_chunk()      # what `slices_in_flight` came to in the timed steps

    # ---- the same K steps again, instrumented: HIP events around every step and call (torch, current stream) and around
    # every stage inside the C-ABI (smx_set_timing(2): events on the stream the kernels are launched on, no
    # synchronisation).  Spreads, the walker's launch duration and the roofline come from this pass.
    events = []
    smx.check(smx.lib().smx_set_timing(2))
    fence()
    t0i = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    fence()
    dti = time.perf_counter() - t0i
    stage = _lib.StageMs()
    stage_rc = smx.lib().smx_stage_times(C.byref(stage))
    smx.check(smx.lib().smx_set_timing(0))
    pipe.check_status()

    def spread(v):
        v = sorted(v)
        return {"median": v[len(v) // 2], "min": v[0], "max": v[-1], "mean": sum(v) / len(v)} if v else None

    agg_ms = [e0.elapsed_time(e1) for _, e0, e1, _ in events]
    step_ms = [es.elapsed_time(ee) for es, _, _, ee in events]
    agg_avg_s = (sum(agg_ms) / max(1, len(agg_ms))) * 1e-3
    cells_per_call = 2.0 * w * h * local_slices   # left + right volume
    operator_achieved = ALGO_BYTES_PER_CELL * cells_per_call / agg_avg_s / 1e9 if agg_avg_s > 0 else 0.0
    # the walker kernel alone: its launches between the stage events of the C-ABI, summed over the timed region
    # (chunked workspaces: several launches per call, each over its share of the cells)
    ncalls = max(1, stage.calls) if stage_rc == 0 else 0
    if stage_rc == 0 and (stage.calls != args.steps or stage.dropped):
        # (the C-ABI keeps at most 32768 stage marks: a longer instrumented pass would average a truncated set)
        print(f"bench.py: stage events of {stage.calls} calls for {args.steps} steps, {stage.dropped} marks dropped -- stage "
              "times cover only the recorded ones", file=sys.stderr)
    walk_avg_s = stage.aggregation / ncalls * 1e-3 if ncalls else 0.0
    achieved = ALGO_BYTES_PER_CELL * cells_per_call / walk_avg_s / 1e9 if walk_avg_s > 0 else operator_achieved

    # HBM bytes of one aggregation call from the committed PMC profile of this same command
    # (tools/pmc.sh + tools/traffic.py; FETCH_SIZE/WRITE_SIZE in separate passes, gfx950 x2 fetch
    # correction calibrated on a kernel with a known byte count).  Only valid for the profiled config.
    traffic = None
    traffic_src = None
    tname = {"kitti": "r05_traffic.json", "motorcycle": "r05_motorcycle_traffic.json",
             "4k": "r05_4k_traffic.json"}.get(args.workload) if args.mode == "exact" else None
    tpath = os.path.join(ROOT, "profiles", tname) if tname else None
    if world == 1 and tpath and args.slices_in_flight is None and args.source == "images" and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            # the profile belongs to one build of the library: a stale file is not quoted
            if tj.get("library") != smx.lib().smx_version().decode():
                raise KeyError("library")
            traffic = float(tj["walker_hbm_bytes"])
            traffic_src = (f"profiles/{tname}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command in "
                           "separate passes, (2*FETCH_SIZE + WRITE_SIZE)*1024 of the walker kernel per launch "
                           "(tools/gpu_suite.sh, tools/traffic.py); an UPPER bound: the x2 FETCH_SIZE rule of the "
                           "guide over-counts narrow loads")
        except (ValueError, KeyError):
            traffic = None

    # VALU issue capacity the walker launch used: its dynamic instruction count (committed PMC profile of this build and
    # workload) x the SIMD cycles of its instruction mix / (LIVE launch duration x clock x SIMDs)  -- tools/valu.py
    valu_frac = None
    vpath = os.path.join(ROOT, "profiles", "r05_valu.json")
    if world == 1 and args.workload == "kitti" and args.mode == "exact" and args.source == "images" and walk_avg_s > 0 \
            and os.path.exists(vpath):
        try:
            vj = json.load(open(vpath))
            if vj.get("library") == smx.lib().smx_version().decode():
                valu_frac = (float(vj["walker_valu_instructions_per_launch"]) * float(vj["simd_cycles_per_valu_instruction"])
                             / (walk_avg_s * float(vj["shader_clock_ghz"]) * 1e9 * float(vj["simds"])))
        except (ValueError, KeyError):
            valu_frac = None

    result = {
        "metric": "disparity MPix/s (stereo pair -> L+R disparity + occlusion-filled map)"
                  + ("" if args.mode == "exact" else " -- FAST mode, NOT bit-exact, not the headline")
                  + ("" if args.source == "images" else " -- from materialised cost volumes (read p + write q), not the headline"),
        "source": args.source,
        "mode": args.mode,
        "value": (w * h * args.steps) / dt / 1e6,
        "unit": "MPix/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "preheat_steps": preheat,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload} {w}x{h} D={D} seeded synthetic pair (seed {seed})",
                   "width": w, "height": h, "disparities": D,
                   "sharding": f"disparity slices / {world} ranks" if world > 1 else "none",
                   "backend": args.backend if world > 1 else None,
                   "slices_in_flight": chunk, "walker_launches_per_call": walker_launches,
                   "library": smx.lib().smx_version().decode()},
        "instrumented_pass": {
            "what": "the same K steps run a second time with HIP events around every step, call and stage; `step_ms`, "
                    "`roofline.avg_launch_ms`, `stage_ms_per_call` and `operator.call_ms` are from this pass (events cost "
                    "bubbles, so its steps are slower than the timed region's)",
            "ms_per_step": dti / args.steps * 1e3,
        },
        "step_ms": spread(step_ms),
        "roofline": {
            "bound": "hbm",
            "scope": "walker_kernel",     # frac / achieved / avg_launch_ms: the dominant KERNEL (rounds 1-3 quoted the whole call: `operator`)
            "kernel": "fused guided-filter aggregation walker, both views per launch (k_v5_walk; k_v4_walk where the comb "
                      "walker does not apply)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            # which wall the kernel is nearer to: share of the chip's VALU issue capacity the launch used (null where no PMC
            # profile of this build and workload is committed).  The kernel is bound by neither: dependent chains per wave.
            "valu_frac": valu_frac,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "avg_launch_ms": walk_avg_s * 1e3,
            "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CELL * cells_per_call,
            "stage_ms_per_call": ({k: getattr(stage, k) / ncalls for k in ("guidance", "aggregation", "wta", "finish")}
                                  if ncalls else None),
            "operator": {
                "what": "the whole smx_dev_aggregate_wta_pair call (guidance statistics, walker, WTA pass): the `frac` of "
                        "the round 1-3 lines",
                "call_ms": spread(agg_ms),
                "achieved": operator_achieved,
                "frac": operator_achieved / HBM_PEAK_GBS,
            },
        },
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle
        oracle.build()
        try:
            os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
        except (AttributeError, OSError):
            pass
        # bounded sample: the full pair when it takes <~30 s at one thread, else a slice subset
        cells = float(w) * h * D * 2
        sample_d = D if cells <= 4e8 else max(4, int(4e8 / (2.0 * w * h)))
        # repeat until ~10 s of CPU work have been timed (bounded sample), report the mean
        reps, total_t = 0, 0.0
        while total_t < 10.0 and reps < 16:
            t0 = time.perf_counter()
            oracle.stereo_pair(Il, Ir, sample_d)
            total_t += time.perf_counter() - t0
            reps += 1
        cdt = total_t / reps
        scale = D / sample_d
        cpu_model, cpu_total = "unknown", os.cpu_count()
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        result["cpu_baseline"] = {
            "value": (w * h) / (cdt * scale) / 1e6,
            "unit": "MPix/s",
            "cores": 1,
            "host_cpu": cpu_model,
            "host_logical_cpus": cpu_total,
            "kind": "port",
            "sample": (f"oracle/smx_oracle.c (gcc -O2 -ffp-contract=off), 1 thread, same pair, "
                       f"{sample_d} of {D} disparities per view"
                       + ("" if sample_d == D else f", time scaled x{scale:.2f} (linear in D)")
                       + f", mean of {reps} runs, {cdt:.2f} s each"),
        }

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
