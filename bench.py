#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X hot path (chaining + gap-fill / split extension) on
synthetic long reads against a GRCh37-sized stand-in reference.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (lamsa_hp_run_uploaded: kernels + result download) over one
batch of reads that is already resident in HBM (reference, reads and seed hits uploaded before the
timed region).  N > 1: launched by torch.distributed.run, one process per GPU, every rank owns its
own shard of the read stream (different reads, same reference) -- weak scaling, no collective on
the data path; only the barrier + MAX-over-ranks of the timing use RCCL.

The JSON line also carries
  roofline     -- algorithmic HBM bytes per launch (B_read = L + 20 H + 4 Cs + ceil(T/4) + 4 Co + 32 Rn summed
                  over the batch) / kernel time measured with HIP events on the kernel's stream
  cpu_baseline -- the oracle (plain-C port of the reference's path) on this box's host cores over a bounded
                  sample of the same batch; a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ["LAMSA_NO_BUILD"] = "1"      # only prebuilt libraries (built by __graft_entry__.build()): never start make / gcc from here
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # BASELINE.json configs; the metric ("10 kbp reads vs GRCh37") is quoted on the ONT config
    "ont10k": dict(read_type="ont2d", profile="ont2d", length=10000, over={}, desc="10 kbp ONT-error reads (12%: sub/ins/del 4% each), -T ont2d"),
    "pb5k": dict(read_type="pacbio", profile="pacbio", length=5000, over={}, desc="5 kbp PacBio-error reads (~15%), -T pacbio"),
    "pb20k": dict(read_type="pacbio", profile="pb20k", length=20000, over={"band_w": 200}, desc="20 kbp reads at 15% error, -T pacbio -w 200"),
    "mol5k": dict(read_type="default", profile="default", length=5000, over={}, desc="5 kbp 1%-error reads, default type"),
    "sv10k": dict(read_type="default", profile="sv10k", length=10000, over={"SV_len_thd": 10000}, desc="10 kbp 1%-error reads, 2/3 with a 1-10 kbp deletion or 1-5 kbp insertion, -V 10000"),
}


def algorithmic_bytes(B, streams, tbases):
    """B_read = L + 20*H + 4*Cs + ceil(T/4) + 4*Co + 32*Rn per read, summed."""
    L = int(B.read_off[-1]); H = int(B.hit_off[-1]); Cs = int(B.h_cig_n[:H].sum())
    T4 = int(np.ceil(tbases.astype(np.int64) / 4.0).sum())
    Co = Rn = 0
    for s in streams:
        i, n_lines = 3, s[1] + s[2]
        for _ in range(n_lines):
            n_res = s[i + 3]; i += 4
            for _ in range(n_res):
                cn = s[i + 6]; Co += cn; Rn += 1; i += 7 + cn
    return L + 20 * H + 4 * Cs + T4 + 4 * Co + 32 * Rn, dict(L=L, H=H, Cs=Cs, T=int(tbases.astype(np.int64).sum()), Co=Co, Rn=Rn)


def count_mapped_bases(B, streams):
    """Sum of the lengths of reads with at least one mapped record (the 'aligned' of aligned Gbase/s)."""
    tot = 0
    for r, s in enumerate(streams):
        i, ok = 3, False
        for _ in range(s[1] + s[2]):
            n_res = s[i + 3]; i += 4
            ok = ok or n_res > 0
            for _ in range(n_res):
                i += 7 + s[i + 6]
        if ok:
            tot += int(B.read_off[r + 1] - B.read_off[r])
    return tot


LAUNCH_KEYS = ["retry", "chain1", "fill1", "chain2", "fill2", "publish", "drain_chain1", "drain_fill1", "drain_chain2", "drain_fill2", "lines_round1", "lines_round2", "dp1_within_fill1",
               "list1_within_fill1", "wave_dp1_within_fill1", "wave_jobs", "wave_jobs_alg_MB", "lane_jobs", "cigars_ahead_MB", "wave_jobs_Mcells"]
N_LAUNCH = len(LAUNCH_KEYS) + 1


def measured_profile(workload, reads):
    """Per-kernel figures from the committed rocprofv3 passes of this workload and batch size (profiles/r*_<workload>_pmc.json, written
    by tools/summarize_prof.py from separate --pmc passes of this same command): HBM bytes per step (FETCH_SIZE doubled per the gfx950
    correction of MI355X_MICROARCH.md + WRITE_SIZE = upper bound), VALU issue fraction.  bench.py cannot collect counters itself; empty
    when no profile of this workload and batch size is committed."""
    import glob
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == workload and d.get("reads_per_step") == reads and "kernels" in d:
            out = {k: v for k, v in d["kernels"].items()}
            out["_source"] = "profiles/" + os.path.basename(f)
            out["_step_upper"] = d.get("hbm_bytes_per_step_upper"); out["_step_lower"] = d.get("hbm_bytes_per_step_lower")
    return out


def measured_traffic(workload, reads):
    """HBM bytes per launch of k_align_batch from the committed PMC passes (profiles/*_final_pmc.json: FETCH_SIZE doubled
    per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE, separate rocprofv3 --pmc passes of this same command).
    bench.py cannot collect counters itself; None when no profile of this workload and batch size is committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_final_pmc.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == workload and d.get("reads_per_step") == reads and "hbm_bytes_per_dispatch_upper" in d:
            best = (float(d["hbm_bytes_per_dispatch_upper"]), "profiles/" + os.path.basename(f))
    return best if best else (None, None)


def shard_seed(rank):
    """Every rank simulates its own shard of the read stream (same reference, different reads)."""
    return 1000 + rank


def reduce_job(dt, totals, world, device=None):
    """Whole-job figures from per-rank ones: MAX of the timed region, SUM of the per-rank counters.
    The only collectives of the bench -- the data path has none (reads are independent)."""
    if world <= 1:
        return float(dt), np.asarray(totals, dtype=np.float64)
    import torch
    import torch.distributed as dist
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor(np.asarray(totals, dtype=np.float64), dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), c.cpu().numpy()


def job_rates(dt, totals, steps):
    """(reads/s, aligned Gbase/s) of the whole job; totals = [reads, mapped bases, bases, failed reads] per step, summed over ranks."""
    return totals[0] * steps / dt, totals[1] * steps / dt / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="ont10k", choices=sorted(WORKLOADS))
    ap.add_argument("--reads", type=int, default=65536, help="reads per batch (= per step) and per GPU; the metric's configuration is 1 M reads over 8 GPUs = 125 k per GPU, "
                    "and one batch holds at most 2^31 seed CIGAR words (~ 125 k reads of this kind)")
    ap.add_argument("--ref-bp", type=int, default=3_100_000_000, help="size of the reference stand-in")
    ap.add_argument("--threads", type=int, default=0, help="host threads for input generation and the CPU baseline (0 = every core of the box, os.cpu_count())")
    ap.add_argument("--stagger-ms", type=float, default=0.0, help="diagnostic: delay between the first two queued steps of a run")
    ap.add_argument("--sequential", action="store_true", help="wait for every step before starting the next (default: consecutive steps are queued two deep)")
    ap.add_argument("--stream-chunks", type=int, default=16, help="chunks pushed through the streaming boundary for the PCIe-inclusive rate, after the timed region (0: skip; "
                    "profiles use 0 so that every k_align_batch dispatch of the run is a resident-batch step)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="target duration of the CPU baseline sample (0: skip)")
    ap.add_argument("--no-repeats", action="store_true")
    ap.add_argument("--default-run-reads", type=int, default=768, help="reads of the default-run check after the CPU baseline: `lamsa aln` WITHOUT -R 0 (stage 4, the BWT rescue, on) of the compiled "
                    "reference and of the product binary on a smaller stand-in whose .bwt / .sa `lamsa index --from-pac` builds on the spot (0: skip)")
    ap.add_argument("--default-run-ref-bp", type=int, default=100_000_000, help="size of that stand-in (the FM index of the 3.1 Gbp one takes a quarter of an hour to build)")
    ap.add_argument("--bare", action="store_true", help="profiling runs: only set-up, warm-up and the timed steps (no one-call / streamed / CPU legs afterwards)")
    ap.add_argument("--rehearse", action="store_true", help="development only: run the N > 1 code path with the gloo backend and every rank on GPU 0 "
                    "(a one-GPU box cannot run RCCL with two ranks); the driver never passes this")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.rehearse:
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = local if world > 1 else 0
    red_dev = None if (world == 1 or a.rehearse) else "cuda"

    import simbatch
    import reflib
    from lamsa_amd import hp

    wl = WORKLOADS[a.workload]
    ncpu = os.cpu_count() or 1
    try:
        ncpu = len(os.sched_getaffinity(0))               # the cores this process may actually use
    except (AttributeError, OSError):
        pass
    threads = ncpu if a.threads <= 0 else max(1, min(a.threads, ncpu))
    if world > 1:
        threads = max(1, threads // world)                 # ranks share the box's cores while they simulate their shards
    t0 = time.time()
    ref = simbatch.SimRef(a.ref_bp, n_contigs=24, seed=5, threads=threads, repeats=not a.no_repeats)
    t_ref = time.time() - t0
    t0 = time.time()
    B = simbatch.SimBatch(ref, a.reads, wl["length"], wl["profile"], seed=shard_seed(rank), threads=threads)
    t_gen = time.time() - t0

    P = hp.make_para(wl["read_type"], **wl["over"])
    h = hp.LamsaHp(P, ref=(ref.pac, ref.l_pac, ref.seq_off, ref.seq_len), device=device)
    Bc = hp.compact_batch(B)                            # the boundary's compact form: one byte per seed-CIGAR element, no offsets
    h.upload_batch(Bc)                                  # inputs resident in HBM before the timed region

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()

    def run_steps(k):
        """k passes over the resident batch, results fetched into host memory every time.  Consecutive passes are queued
        two deep (lamsa_hp_start_uploaded / finish_uploaded) like the kernels of consecutive training steps: the waves
        of pass i+1 start on the SIMDs that the tail of pass i leaves idle.  --sequential waits for every pass instead."""
        ms, raw = [], None

        def note():
            ms.append(h.last_kernel_ms(0) + h.last_kernel_ms(1))
            phase_ms.append([h.last_kernel_ms(i) for i in range(1, N_LAUNCH)])
        if a.sequential or k < 2:
            for _ in range(k):
                raw = h.run_uploaded(fetch=True, raw=True)      # kernels + download of the result streams into host memory
                note()
            return ms, raw
        h.start_uploaded()
        if a.stagger_ms > 0:
            time.sleep(a.stagger_ms / 1000.0)           # diagnostic: the second batch enters the pipeline this much later than the first
        for _ in range(k - 1):
            h.start_uploaded()
            raw = h.finish_uploaded(fetch=True, raw=True)
            note()
        raw = h.finish_uploaded(fetch=True, raw=True)
        note()
        return ms, raw

    phase_ms = []                                       # per step: retry pass, then the five launches of the main pass (HIP events)
    if not a.sequential:                                # set-up, untimed: both launch lanes allocate their scratch and output buffers
        h.start_uploaded(); h.start_uploaded(); h.finish_uploaded(fetch=False); h.finish_uploaded(fetch=False)
    if a.warmup:
        run_steps(a.warmup)
    sync()
    del phase_ms[:]
    t0 = time.perf_counter()
    kernel_ms, raw = run_steps(a.steps)
    sync()
    dt = time.perf_counter() - t0
    stream, r_off, r_len, status = raw
    tbases = np.array(h.last_tbases, copy=True); status = np.array(status, copy=True)
    streams = [stream[int(r_off[i]):int(r_off[i]) + int(r_len[i])].tolist() for i in range(a.reads)]   # untimed: parsing for the report
    n_bases = int(B.read_off[-1])
    mapped_bases = count_mapped_bases(B, streams)
    n_fail = int((status != 0).sum())
    dt, tot = reduce_job(dt, [a.reads, mapped_bases, n_bases, n_fail], world, device=red_dev)   # whole-job totals over ranks
    reads_per_s, gbase_per_s = job_rates(dt, tot, a.steps)

    if rank == 0:
        alg_bytes, parts = algorithmic_bytes(B, streams, tbases)
        lm = dict(zip(LAUNCH_KEYS, [float(x) for x in np.mean(np.array(phase_ms), axis=0)]))
        work = np.array(h.last_work, dtype=np.int64).reshape(-1, 4) if len(h.last_work) else np.zeros((1, 4), np.int64)
        cells, pairs, cs_lines = int(work[:, 0].sum()), int(work[:, 1].sum()), int(work[:, 2].sum())
        k_ms = float(np.mean(kernel_ms))
        # The hot path is a sequence of launches per step (hp_phase.h); algorithmic bytes by what each launch group touches, every byte once:
        # chaining reads the hit records (20 B per hit); the fill group (job listing, lane DP, fill) reads the read bases, the 2-bit
        # reference windows of its DP jobs and the seed CIGARs of the hits that ended up on lines (counted by the kernels, like T),
        # and writes the records and their CIGARs.  The seed CIGARs of all the other hits are only touched when the batch is made
        # resident (k_cig8_expand, outside the timed step): `batch_setup` below, not a launch of the step.
        fill_ms = lm["fill1"] + lm["fill2"]
        fill_bytes = parts["L"] + int(np.ceil(tbases.astype(np.int64) / 4.0).sum()) + 4 * cs_lines + 4 * parts["Co"] + 32 * parts["Rn"]
        chain_bytes = 20 * parts["H"]
        parts["Cs_on_lines"] = cs_lines
        step_bytes = chain_bytes + fill_bytes
        groups = [("k_chain1+k_chain2", lm["chain1"] + lm["chain2"], chain_bytes),
                  ("k_filllist+k_filldp_wave+k_filldp_small+k_fill", fill_ms, fill_bytes)]
        # the dominant KERNEL of the step: the longest single launch (round-1 dispatches; round 2 is nearly empty on these workloads)
        per_kernel = lambda m: {"k_chain1": m["chain1"], "k_fill": m["fill1"] - m["dp1_within_fill1"], "k_filldp_wave": m["wave_dp1_within_fill1"],
                                "k_filldp_small": m["dp1_within_fill1"] - m["list1_within_fill1"] - m["wave_dp1_within_fill1"], "k_filllist": m["list1_within_fill1"]}
        dom_kernel = max(per_kernel(lm).items(), key=lambda kv: kv[1])[0]
        pick = lambda m: per_kernel(m)[dom_kernel]
        dom_ms_timed = pick(lm)
        # its algorithmic bytes: chaining = the hit records; the wave-per-job DP = the sequences its jobs read and the CIGARs they write (counted
        # by the kernel); the fill = the fill group's bytes less those
        wave_bytes = int(lm["wave_jobs_alg_MB"] * 1e6)
        dom_bytes = {"k_chain1": chain_bytes, "k_filldp_wave": wave_bytes}.get(dom_kernel, max(fill_bytes - wave_bytes, 0))
        # The launches of the timed region overlap (steps queued two deep): a clean duration of the dominant kernel comes from two
        # further steps run one at a time, with the same HIP events -- the figure a rocprofv3 --kernel-trace average agrees with.
        lm_seq = None
        if not a.sequential and not a.bare:
            del phase_ms[:]
            for _ in range(2):
                h.run_uploaded(fetch=True, raw=True)
                phase_ms.append([h.last_kernel_ms(i) for i in range(1, N_LAUNCH)])
            lm_seq = dict(zip(LAUNCH_KEYS, [float(x) for x in np.mean(np.array(phase_ms), axis=0)]))
        dom_ms = pick(lm_seq) if lm_seq else dom_ms_timed
        achieved = dom_bytes / max(dom_ms, 1e-6) / 1e6
        prof = measured_profile(a.workload, a.reads)
        pk = lambda k, c: (prof.get(k, {}).get("per_dispatch", {}) or {}).get(c)
        chain_traffic = prof.get("k_chain1", {}).get("hbm_bytes_per_step_upper")
        valu_fill = sum((pk(k, "SQ_INSTS_VALU") or 0) * (prof.get(k, {}).get("dispatches_per_step") or 1) for k in ("k_fill", "k_filldp_small", "k_filldp_wave", "k_filllist"))
        roof = {"bound": "hbm", "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 6),
                "traffic": prof.get(dom_kernel, {}).get("hbm_bytes_per_step_upper"), "traffic_source": prof.get("_source"),
                "kernel": dom_kernel, "kernel_ms": round(dom_ms, 3), "kernel_ms_timed_region_overlapped": round(dom_ms_timed, 3),
                "kernel_ms_rocprof_avg_committed": (prof.get(dom_kernel, {}).get("round1") or {}).get("avg_ms_rocprof") or prof.get(dom_kernel, {}).get("ms_per_step_rocprof"),
                "issue_busy_committed": {k: {"valu": (v.get("round1") or v).get("valu_busy_frac"), "scalar": (v.get("round1") or v).get("salu_busy_frac")} for k, v in prof.items() if isinstance(v, dict) and (v.get("round1") or v).get("valu_busy_frac") is not None} or None,
                "algorithmic_bytes_per_launch": int(dom_bytes),
                "note": "dominant launch of the step; duration = HIP events on the launches' own stream, %s; its algorithmic bytes = the terms of B_read that launch group touches "
                        "(seed-CIGAR words only of the hits on lines, counted by the kernel)" % ("steps one at a time after the timed region (the timed region queues steps two deep, so its event intervals overlap)" if lm_seq else "timed region"),
                "whole_step": {"algorithmic_bytes": int(step_bytes), "launch_ms_sum": round(k_ms, 3), "achieved_GBps": round(step_bytes / max(k_ms, 1e-6) / 1e6, 3),
                               "hbm_traffic_bytes_upper": prof.get("_step_upper"), "hbm_traffic_bytes_lower": prof.get("_step_lower"),
                               "b_read_formula_bytes": int(alg_bytes)},
                "batch_setup": {"what": "k_cig8_expand + offset scan when the batch is made resident, outside the timed step", "algorithmic_bytes": int(5 * parts["Cs"])},
                "launch_groups": [{"launches": g[0], "ms": round(g[1], 3), "algorithmic_bytes": int(g[2]), "achieved_GBps": round(g[2] / max(g[1], 1e-6) / 1e6, 3)} for g in groups],
                "launch_ms_one_step_at_a_time": {k: round(v, 3) for k, v in lm_seq.items()} if lm_seq else None,
                "bytes_per_read": round(step_bytes / a.reads, 1), "terms": parts,
                # the compute side: DP cell updates and chaining edge classifications actually executed
                "dp_cells_per_step": cells, "gcups": round(cells * a.steps * world / dt / 1e9, 3), "gcups_within_fill_launches": round(cells / max((lm_seq["fill1"] + lm_seq["fill2"]) if lm_seq else fill_ms, 1e-6) / 1e6, 3),      # (launch durations one step at a time when measured)
                "pair_evals_per_step": pairs, "pair_evals_per_s": round(pairs * a.steps * world / dt, 1), "pair_evals_per_s_within_chain_launches": round(pairs / max((lm_seq or lm)["chain1"] + (lm_seq or lm)["chain2"], 1e-6) * 1e3, 1),
                # per unit of work, from the committed counter passes of this workload (None without one)
                "hbm_bytes_per_pair_eval": round(chain_traffic / max(pairs, 1), 2) if chain_traffic else None,
                "valu_lane_slots_per_cell": round(valu_fill * 64.0 / max(cells, 1), 1) if valu_fill else None,
                "valu_issue_frac": {k: v.get("valu_issue_frac") for k, v in prof.items() if isinstance(v, dict) and v.get("valu_issue_frac") is not None} or None}
        # the compute-side ceiling of the DP launch (the path is max-plus DP, not bytes): how busy the two issue ports were (committed counter pass),
        # what a cell cost in VALU lane-slots, and what the recurrence alone needs (~22 lane operations per cell -- substitution
        # score 3, M 2, E 4, F 4, H and the direction nibble 6, the matrix store 3 -- before any cross-lane work of a row)
        wave = prof.get("k_filldp_wave", {})
        wv_ = wave.get("round1") or wave
        wave_valu = (wave.get("per_dispatch") or {}).get("SQ_INSTS_VALU") or (wv_.get("per_dispatch") or {}).get("SQ_INSTS_VALU")
        wave_cells = lm["wave_jobs_Mcells"] * 1e6
        roof["compute"] = {"kernel": "k_filldp_wave", "valu_busy": wv_.get("valu_busy_frac"), "salu_busy": wv_.get("salu_busy_frac"),
                           "lane_slots_per_cell": round(wave_valu * 64.0 / wave_cells, 1) if (wave_valu and wave_cells) else None,
                           "lane_slots_per_cell_recurrence_only": 22, "dp_cells_per_launch": int(wave_cells),
                           "gcups_of_the_launch": round(wave_cells / max(lm_seq["wave_dp1_within_fill1"] if lm_seq else lm["wave_dp1_within_fill1"], 1e-6) / 1e6, 1) if wave_cells else None,
                           "source": prof.get("_source")}
        t_pcie = r_seq = float("nan")
        if not a.bare:
            # PCIe-inclusive rate of the one-call boundary (host buffers in, host results out), one untimed step; never `value`
            t0 = time.perf_counter(); h.upload_batch(Bc); h.run_uploaded(fetch=True, raw=True); t_pcie = time.perf_counter() - t0
            # the same resident batch one step at a time (every step waited for before the next starts), for comparison with `value`
            t0 = time.perf_counter()
            for _ in range(2):
                h.run_uploaded(fetch=True, raw=True)
            r_seq = 2 * a.reads / (time.perf_counter() - t0)

        def streamed(Bx, k=4):
            """k chunks through the streaming boundary (submit/collect, two in flight): upload of chunk i overlaps the kernel of chunk i-1"""
            t0 = time.perf_counter()
            h.submit_batch(Bx)
            for _ in range(1, k):
                h.submit_batch(Bx); h.collect_batch(raw=True)
            h.collect_batch(raw=True)
            return k * a.reads / (time.perf_counter() - t0)
        r_stream = r_stream_pinned = None
        if a.stream_chunks > 1 and not a.bare:
            r_stream = round(streamed(Bc, a.stream_chunks), 2)
            Bp = hp.pinned_batch(Bc)
            r_stream_pinned = round(streamed(Bp, a.stream_chunks), 2)
            Bp.release()
        cpu = None
        h.close()                                          # every GPU measurement is taken: the device buffers go back before the product binary (a process of its own) is run beside the reference's
        if a.cpu_seconds > 0 and world == 1 and not a.bare:               # the CPU baseline is measured on rank 0 of the single-GPU run only
            lp = reflib.lo_para(wl["read_type"], **wl["over"])
            probe = min(256, a.reads)                      # estimate the rate on a probe, then size the sample for ~cpu_seconds
            # The box shows every core of the host but may grant this job a share of them (a CPU quota): the thread count is chosen by
            # a probe -- all visible cores, 64, 32, 16 -- and the baseline runs with the fastest; `cores` is what it then used.
            cand = sorted({threads} | {t for t in (64, 32, 16) if t < threads}, reverse=True)
            rates = {}
            for t in cand:
                t0 = time.perf_counter(); reflib.oracle_streams(take_first(B, probe), lp, t); rates[t] = probe / max(time.perf_counter() - t0, 1e-3)
            cpu_threads = max(rates, key=lambda t: rates[t]); tp = probe / rates[cpu_threads]
            n_s = int(max(probe, min(a.reads, probe * a.cpu_seconds / max(tp, 1e-3))))
            sample = take_first(B, n_s)
            t0 = time.perf_counter(); want = reflib.oracle_streams(sample, lp, cpu_threads); tc = time.perf_counter() - t0
            same = sum(1 for i in range(n_s) if want[i] == streams[i])
            cpu = {"value": round(float(sample.read_off[-1]) / tc / 1e9, 6), "unit": "Gbase/s", "cores": cpu_threads, "kind": "port",
                   "sample": "first %d reads of the rank-0 batch, oracle (plain-C port of the reference path, one read per task) on %d threads (of %d visible cores; "
                             "probe of %d reads, reads/s by thread count: %s), %.1f s" % (n_s, cpu_threads, threads, probe, ", ".join("%d: %.0f" % (t, rates[t]) for t in cand), tc),
                   "reads_per_s": round(n_s / tc, 3), "gpu_equals_cpu_on_sample": "%d/%d reads" % (same, n_s), "host_cpu": cpu_model()}
            try:
                cpu["reference_binary"] = reference_binary_baseline(B, ref, wl, cpu_threads, a.cpu_seconds)
            except Exception as e:                           # a reported extra, never a reason to lose the bench line
                cpu["reference_binary"] = {"error": repr(e)[:200]}
            if a.default_run_reads > 0 and isinstance(cpu.get("reference_binary"), dict):
                try:                                         # the reference's DEFAULT run (stage 4 on) against the product's, on an index built here
                    cpu["reference_binary"]["default_run"] = default_run_check(a.workload, wl, cpu_threads, a.default_run_reads, a.default_run_ref_bp)
                    cpu["reference_binary"]["gpu_equals_reference_on_sample_default_run"] = cpu["reference_binary"]["default_run"].get("gpu_equals_reference")
                except Exception as e:
                    cpu["reference_binary"]["default_run"] = {"error": repr(e)[:300]}
        hits = np.diff(B.hit_off)
        out = {
            "metric": "aligned Gbase/s, %s vs GRCh37-sized stand-in; inputs resident in HBM when the timed region starts (upload excluded: see pcie_inclusive_*)" % wl["desc"], "value": round(gbase_per_s, 6), "unit": "Gbase/s",
            "reads_per_s": round(reads_per_s, 2),
            "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: %d reads/step/GPU x %d bp; reference stand-in %d bp in 24 contigs, %d repeat copies; seed hits simulated "
                                   "(GEM thresholds, <=200/seed): %.1f hits/seed, %.0f hits/read" % (a.workload, a.reads, wl["length"], ref.l_pac, ref.n_copies, hits.mean() if len(hits) else 0, B.n_hits / max(1, a.reads)),
                       "reads_per_step_per_gpu": a.reads, "read_len": wl["length"], "read_type": wl["read_type"], "parallelism": "reads sharded over %d GPU(s), no collectives" % a.gpus},
            "reads_not_ok": int(tot[3]), "steps_queued_two_deep": not a.sequential, "reads_per_s_one_step_at_a_time_rank0": None if a.bare else round(r_seq, 2),
            "pcie_inclusive_reads_per_s": None if a.bare else round(a.reads / t_pcie, 2),
            "pcie_inclusive_streamed_reads_per_s": {"pageable_host_arrays": r_stream, "pinned_host_arrays": r_stream_pinned, "chunks": a.stream_chunks},
            "setup_s": {"reference": round(t_ref, 1), "reads_and_hits": round(t_gen, 1)},
            "launch_ms": {k: round(v, 3) for k, v in lm.items()},
            "roofline": roof, "cpu_baseline": cpu,
            "library": dict(zip(("path", "sha256"), hp.loaded_library())),
        }
        print(json.dumps(out))
    h.close()
    if world > 1:
        dist.destroy_process_group()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def reference_binary_baseline(B, ref, wl, threads, seconds):
    """`lamsa aln -t <all cores> -N -I -R 0` of the REFERENCE ITSELF (oracle/_ref/lamsa, compiled from /root/reference in the build
    container and shipped as a binary) on files holding a sample of the same batch: FASTA reads, the seed hits as GEM map text,
    .seed.info, and the stand-in's .ann/.amb/.pac.  The reference refuses to start without <ref>.bwt/.sa (src/lamsa_aln.c:1233);
    with -R 0 they are loaded and never searched, so the pair of the small fixture reference (tests/golden/ref) stands in for
    them -- nothing on this box can build the BWT of a 3.1 Gbp text.  Wall time of the whole process (index load, text parse,
    alignment, SAM) as in BASELINE.md's "hot path" column.  None when the binary did not travel."""
    import shutil
    import subprocess
    import tempfile
    import simfiles
    exe = os.path.join(ROOT, "oracle", "_ref", "lamsa")
    if not os.path.exists(exe):
        return None
    p = __import__("simbatch").PROFILES[wl["profile"]]
    args = [] if wl["read_type"] == "default" else ["-T", wl["read_type"]]
    flag = {"band_w": "-w", "SV_len_thd": "-V"}
    for k, v in wl["over"].items():
        args += [flag[k], str(v)]
    d = tempfile.mkdtemp(prefix="lamsa_cpu_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        simfiles.write_index(d + "/ref.fa", ref)
        for ext in ("bwt", "sa"):
            shutil.copy(os.path.join(ROOT, "tests", "golden", "ref", "ref.fa." + ext), d + "/ref.fa." + ext)

        def run(n):
            sample = take_first(B, n)
            simfiles.write_reads(d + "/reads.fa", sample, seed_len=50, seed_step=p["seed_step"], workers=min(threads, 32))
            with open(d + "/reads.fa.seed.info", "w") as f:
                for r in range(n):
                    f.write("r%d %d %d %d\n" % (r, int(sample.seed_all[r]), int(sample.last_len[r]), int(sample.read_off[r + 1] - sample.read_off[r])))
            t0 = time.perf_counter()
            q = subprocess.run([exe, "aln"] + args + ["-t", str(threads), "-N", "-I", "-R", "0", "-o", d + "/out.sam", d + "/ref.fa", d + "/reads.fa"],
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=600)
            dt = time.perf_counter() - t0
            if q.returncode != 0:
                raise RuntimeError("reference exited with %d: %s" % (q.returncode, q.stderr[-200:]))
            return sample, dt
        probe = min(128, B.n_reads)
        _, t_load = run(1)                                  # start-up: .pac / .ann load of the stand-in reference
        _, tp = run(probe)
        rate = probe / max(tp - t_load, 1e-3)
        n = int(max(probe, min(B.n_reads, rate * seconds)))
        sample, dt = run(n)
        out = {"value": round(float(sample.read_off[-1]) / dt / 1e9, 6), "unit": "Gbase/s", "cores": threads, "kind": "reference",
               "reads_per_s": round(n / dt, 3), "reads_per_s_excluding_startup": round(n / max(dt - t_load, 1e-3), 3), "startup_s": round(t_load, 2),
               "sample": "first %d reads as files, `lamsa aln %s -t %d -N -I -R 0` of the compiled reference, %.1f s wall" % (n, " ".join(args), threads, dt)}
        # At-scale parity against the reference ITSELF: the product binary (host C++ over the C-ABI, HIP kernels) on the same files,
        # SAM compared record by record with the reference's (everything but the @PG line).
        out["gpu_equals_reference_on_sample"] = compare_with_product(d, args, threads, n)
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def default_run_check(workload, wl, threads, n_reads, ref_bp, exe=None, keep=None, prebuilt=None, unseeded=None):
    """The reference's DEFAULT run -- no -R 0: stage 4, the BWT rescue of unaligned read parts (src/bwt_aln.c:398-409), searches the FM
    index -- against the product binary's, on files: a stand-in of ref_bp bases with the bench's repeat families, its .bwt / .sa built here by
    the product's own `lamsa index --from-pac`, n_reads simulated reads of the workload with their seed hits as GEM map text.  Returns the
    per-read SAM comparison, both wall times, and what share of the product's chunk loop went to stage 4 (its own trace).  unseeded = (share
    of the reads, share of a read's seeds): those reads get a stretch without any seed hit -- on fully seeded simulated reads stage 4 finds
    nothing to do."""
    import re
    import shutil
    import subprocess
    import tempfile
    import simbatch
    import simfiles
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "lamsa")
    exe = exe or os.path.join(ROOT, "lamsa_amd", "bin", "lamsa")
    if not os.path.exists(ref_exe) or not os.path.exists(exe):
        return {"error": "the compiled reference or the product binary is not on this machine"}
    p = simbatch.PROFILES[wl["profile"]]
    args = [] if wl["read_type"] == "default" else ["-T", wl["read_type"]]
    for k, v in wl["over"].items():
        args += [{"band_w": "-w", "SV_len_thd": "-V"}[k], str(v)]
    d = keep or tempfile.mkdtemp(prefix="lamsa_dflt_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        t0 = time.perf_counter()
        if prebuilt:                                        # (tools/default_run.py: one stand-in and one index for every workload)
            ref, t_index = prebuilt
        else:
            ref = simbatch.SimRef(ref_bp, n_contigs=12, seed=17, threads=min(threads, 32))
            simfiles.write_index(d + "/ref.fa", ref)
            t1 = time.perf_counter()
            q = subprocess.run([exe, "index", "--from-pac", d + "/ref.fa"], capture_output=True, text=True, timeout=1500, env=dict(os.environ, LAMSA_INDEX_THREADS=str(min(threads, 64))))
            if q.returncode != 0:
                return {"error": "lamsa index --from-pac failed: " + q.stderr[-300:]}
            t_index = time.perf_counter() - t1
        B = simbatch.SimBatch(ref, n_reads, wl["length"], wl["profile"], seed=23, threads=min(threads, 32))
        simfiles.write_reads(d + "/reads.fa", B, seed_len=50, seed_step=p["seed_step"], workers=min(threads, 32), unseeded=unseeded)
        with open(d + "/reads.fa.seed.info", "w") as f:
            for r in range(n_reads):
                f.write("r%d %d %d %d\n" % (r, int(B.seed_all[r]), int(B.last_len[r]), int(B.read_off[r + 1] - B.read_off[r])))
        t_setup = time.perf_counter() - t0
        t2 = time.perf_counter()
        q = subprocess.run([ref_exe, "aln"] + args + ["-t", str(threads), "-N", "-I", "-o", d + "/out.sam", d + "/ref.fa", d + "/reads.fa"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=1500)
        t_ref = time.perf_counter() - t2
        if q.returncode != 0:
            return {"error": "reference exited with %d: %s" % (q.returncode, q.stderr[-300:])}
        t3 = time.perf_counter()
        g = subprocess.run([exe, "aln"] + args + ["-t", str(min(threads, 32)), "-N", "-I", "-o", d + "/out_gpu.sam", d + "/ref.fa", d + "/reads.fa"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=1500,
                           env=dict(os.environ, LAMSA_TRACE="1"))
        t_gpu = time.perf_counter() - t3
        if g.returncode != 0:
            return {"error": "product binary exited with %d: %s" % (g.returncode, " | ".join(l for l in g.stderr.splitlines() if "read " not in l)[-400:])}
        same = sam_same_reads(d + "/out.sam", d + "/out_gpu.sam")
        # the product's own account of its write stage: "[write] N reads: records + stage-4 plan A s, stage-4 DP batch (J jobs) B s, finish + rank + SAM C s"
        plan = dp = rest = 0.0; jobs = 0
        for m in re.finditer(r"\[write\] \d+ reads: records(?: \+ stage-4 plan)? ([0-9.]+) s, stage-4 DP batch \((\d+) jobs\) ([0-9.]+) s, finish \+ rank \+ SAM(?: text)? ([0-9.]+) s", g.stderr):
            plan += float(m.group(1)); jobs += int(m.group(2)); dp += float(m.group(3)); rest += float(m.group(4))
        n_rescued = sum(1 for line in open(d + "/out.sam") if not line.startswith("@"))
        return {"workload": workload, "reads": n_reads, "unseeded_stretch": unseeded, "ref_bp": int(ref.l_pac), "gpu_equals_reference": same, "sam_lines_reference": n_rescued,
                "index_build_s": round(t_index, 1), "setup_s": round(t_setup, 1), "reference_wall_s": round(t_ref, 2), "reference_threads": threads, "product_wall_s": round(t_gpu, 2),
                "product_stage4": {"dp_jobs": jobs, "plan_s": round(plan, 3), "dp_batch_s": round(dp, 3), "finish_rank_sam_s": round(rest, 3),
                                   "share_of_product_wall": round((plan + dp) / max(t_gpu, 1e-6), 4)},
                "command": "lamsa aln %s -N -I  (no -R: stage 4 on)" % " ".join(args)}
    finally:
        if not keep:
            shutil.rmtree(d, ignore_errors=True)


def sam_same_reads(a, b):
    def by_read(path):
        recs, hdr = {}, []
        for line in open(path):
            if line.startswith("@"):
                if not line.startswith("@PG"):
                    hdr.append(line)
                continue
            recs.setdefault(line.split("\t", 1)[0], []).append(line)
        return hdr, recs
    h1, r1 = by_read(a); h2, r2 = by_read(b)
    same = sum(1 for k in r1 if r2.get(k) == r1[k])
    note = "" if (h1 == h2 and len(r1) == len(r2)) else " (headers or read sets differ: %d vs %d reads)" % (len(r1), len(r2))
    return "%d/%d reads%s" % (same, len(r1), note)


def compare_with_product(d, args, threads, n, exe=None):
    """`lamsa_amd/bin/lamsa aln <args> -N -I -R 0` on the files in d (ref.fa.*, reads.fa, reads.fa.seed.gem.map) against d/out.sam, the
    reference's own output on them.  Returns "<identical reads>/<reads>" (a read = all SAM lines with its name) or an error note.
    exe: another build of the host program (the CPU tests pass the one linked against the emulated C-ABI)."""
    import subprocess
    exe = exe or os.path.join(ROOT, "lamsa_amd", "bin", "lamsa")
    if not os.path.exists(exe):
        return "product binary not built"
    q = subprocess.run([exe, "aln"] + args + ["-t", str(min(threads, 32)), "-N", "-I", "-R", "0", "-o", d + "/out_gpu.sam", d + "/ref.fa", d + "/reads.fa"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=900)
    if q.returncode != 0:
        return "product binary exited with %d: %s" % (q.returncode, " | ".join(l for l in q.stderr.splitlines() if "read " not in l)[-600:])

    def by_read(path):
        recs, hdr = {}, []
        for line in open(path):
            if line.startswith("@"):
                if not line.startswith("@PG"):
                    hdr.append(line)
                continue
            recs.setdefault(line.split("\t", 1)[0], []).append(line)
        return hdr, recs
    h1, r1 = by_read(d + "/out.sam"); h2, r2 = by_read(d + "/out_gpu.sam")
    same = sum(1 for k in r1 if r2.get(k) == r1[k])
    note = "" if (h1 == h2 and len(r1) == len(r2)) else " (headers or read sets differ: %d vs %d reads)" % (len(r1), len(r2))
    return "%d/%d reads%s" % (same, len(r1), note)


def take_first(B, n):
    """First n reads of a batch as a new batch object (numpy slices; offsets stay valid because they start at 0)."""
    class _S:
        pass
    s = _S()
    ns = int(B.seed_off[n]); nh = int(B.hit_off[ns])
    s.n_reads = n
    s.read_off = B.read_off[:n + 1].copy(); s.read_seq = B.read_seq[:max(int(B.read_off[n]), 1) + 4].copy()
    s.seed_all = B.seed_all[:max(n, 1)].copy(); s.last_len = B.last_len[:max(n, 1)].copy(); s.seed_off = B.seed_off[:n + 1].copy()
    s.seed_id = B.seed_id[:max(ns, 1)].copy(); s.hit_off = B.hit_off[:ns + 1].copy()
    for k in ("h_pos", "h_chr", "h_strand", "h_nm", "h_len_dif", "h_cig_off", "h_cig_n"):
        setattr(s, k, getattr(B, k)[:max(nh, 1)].copy())
    s.cig = B.cig
    s.pac, s.l_pac, s.seq_off, s.seq_len = B.pac, B.l_pac, B.seq_off, B.seq_len
    return s


if __name__ == "__main__":
    main()
