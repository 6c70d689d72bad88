"""Turn rocprofv3 (ROCm 7.2, rocpd sqlite output) result DBs under gpurun_out/ into the committed text / JSON summaries.

  python profiles/summarize.py r01        # reads gpurun_out/r01_trace, r01_pmc_*  ->  profiles/r01_*.txt|json

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in SEPARATE --pmc
passes, are reported in KiB, and on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams, so the read
side is doubled ("fetch_x2"); WRITE_SIZE is taken as reported (uncalibrated).
"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def db(path_glob):
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", path_glob, "*", "*_results.db")))
    return sqlite3.connect(f[-1]) if f else None


def main(tag):
    out_txt = []
    con = db(f"{tag}_trace")
    if con:
        out_txt.append(f"# rocprofv3 --kernel-trace --stats  ({tag})   name | calls | total_us | avg_us | pct")
        for r in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            out_txt.append(f"{r[0][:100]:100s} {r[1]:6d} {r[2]:14.1f} {r[3]:12.2f} {r[4]:6.2f}")
        out_txt.append("")
        out_txt.append("# per-kernel resources: name | vgpr | accum_vgpr | sgpr | lds | scratch | grid | wg")
        seen = set()
        for r in con.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x from kernels"):
            if r[0] in seen:
                continue
            seen.add(r[0])
            out_txt.append(" | ".join(str(x) for x in r)[:200])
    pmc = {}
    for kind in ("fetch", "write", "sq", "sq2", "sq3", "tcp", "ta"):
        con = db(f"{tag}_pmc_{kind}")
        if not con:
            continue
        q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
             "group by kernel_name, counter_name")
        for name, cname, n, avg, tot in con.execute(q):
            pmc.setdefault(name, {})[cname] = {"launches": n, "avg": avg, "sum": tot}
    summary = {"tag": tag, "per_kernel": pmc}
    # the kernel of bench.py's TIMED region: the sampler's table-reading instantiation (`..., false, 1>`) when it ran, else the per-edge one
    # (which also serves the untimed score-forward figure of the same command)
    edge_names = [n for n in pmc if n.startswith("void k_edge")]
    def mode_of(name):          # k_edge<L, F0, HP, H1, H2, UN, MODE[, NW]>: template argument 6
        args = name.split("<", 1)[1].rsplit(">", 1)[0].split(",") if "<" in name else []
        return args[6].strip() if len(args) > 6 else "0"
    def is_unet(name):
        args = name.split("<", 1)[1].rsplit(">", 1)[0].split(",") if "<" in name else []
        return len(args) > 5 and args[5].strip() == "true"
    timed = [n for n in edge_names if mode_of(n) == "1"] or [n for n in edge_names if not is_unet(n)] or edge_names
    summary["edge_kernel_name"] = timed[-1] if timed else None
    # average duration of the timed kernel in the kernel trace of the same command, and the MFMA issue fraction that follows from it:
    # SQ_INSTS_MFMA x 32 768 FLOP (one v_mfma_f32_32x32x16_f16) / launch time / 2.5 PFLOP/s dense fp16
    con = db(f"{tag}_trace")
    if con and timed:
        for r in con.execute("select name, average from top_kernels"):
            if r[0] == timed[-1]:
                summary["edge_kernel_avg_us_in_trace"] = r[1]
        # Steady state of the timed instantiation: the all-launch average above contains the COLD first launch (code object upload, cold L2 /
        # instruction cache: ~+10 % on a 5-step trace), which bench.py's HIP-event average over its timed region does not.  From the
        # per-dispatch rows: the same average without the first launch, and the median.
        try:
            cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
            if "start" in cols and "end" in cols:
                rows = {}
                for name, t0, t1 in con.execute("select name, start, end from kernels order by start"):
                    rows.setdefault(name, []).append((t1 - t0) * 1e-3)          # ns -> us
                d = rows.get(timed[-1], [])
                if len(d) >= 2:
                    st = d[1:]
                    summary["edge_kernel_launches_in_trace"] = len(d)
                    summary["edge_kernel_first_launch_us"] = d[0]
                    summary["edge_kernel_steady_avg_us_in_trace"] = sum(st) / len(st)
                    summary["edge_kernel_median_us_in_trace"] = sorted(d)[len(d) // 2]
                # what bench.py's "edge" class also brackets per step: the table generator (and, in dedf_score, its check) of the same shape
                for name, dd in rows.items():
                    if name.startswith("void k_radial_table") and len(dd) >= 2:
                        summary.setdefault("radial_table_steady_avg_us_in_trace", {})[name[:60]] = sum(dd[1:]) / len(dd[1:])
        except sqlite3.Error as e:          # (schema of another rocprofv3 version: keep the all-launch figures)
            summary["steady_state_error"] = str(e)
        # the bench line of the traced command (edges per launch, workload) -> the executed-FLOP fraction recomputed from THIS trace
        try:
            line = open(os.path.join(ROOT, "gpurun_out", f"{tag}_trace_bench.json")).read().strip().splitlines()[-1]
            b = json.loads(line)
            edges = b["config"]["edges_per_step_rank0"]
            lmax = 3 if "lmax=3" in b["metric"] else (1 if "lmax=1" in b["metric"] else 2)
            table_on = b["roofline"]["radial_table"].get("enabled", False)
            m_edge = {1: 105_536, 2: 193_344, 3: 338_304}[lmax] - (40_960 if table_on else 0)
            us = summary.get("edge_kernel_steady_avg_us_in_trace")
            # the bench line's edge count is that of its TIMED steps: compare with the launches of those steps only (the warm-up steps before
            # them start from the seeded poses and carry a different number of edges)
            try:
                d_all = rows.get(timed[-1], [])
                k = int(b.get("steps", 0))
                if k and len(d_all) >= k:
                    us = sum(d_all[-k:]) / k
                    summary["edge_kernel_timed_steps_avg_us_in_trace"] = us
            except Exception:          # noqa: BLE001
                pass
            if us:
                gen = sum(summary.get("radial_table_steady_avg_us_in_trace", {}).values()) if table_on else 0.0
                m_alg = {1: 105_536, 2: 193_344, 3: 338_304}[lmax]
                # roofline.frac of the bench line (round 6: ONE definition, SURVEY 8(d)'s algorithmic constant) recomputed from THIS trace
                summary["edge_kernel_frac_from_trace"] = 2.0 * m_alg * edges / ((us + gen) * 1e-6) / (2.5e15 / 3.0)
                summary["edge_kernel_frac_from_trace_definition"] = ("2 x algorithmic MAC per edge (SURVEY 8(d)) x edges per launch of the traced bench line / (average duration of the "
                                                                     "timed kernel over the timed steps + its table generator) / (2.5 PFLOP/s / 3)")
                summary["traced_bench_line_frac"] = b["roofline"]["frac"]
                summary["edge_kernel_frac_round2_4_units_from_trace"] = 2.0 * m_edge * edges / ((us + gen) * 1e-6) / (2.5e15 / 3.0)
                summary["traced_bench_line_frac_round2_4_units"] = b["roofline"].get("frac_round2_4_units")
                summary["traced_bench_line_edges_per_step"] = edges
                summary["traced_bench_line_gemm_mac_per_edge_executed"] = b["roofline"].get("gemm_mac_per_edge_executed")
                summary["traced_bench_line_frac_gemm_executed"] = b["roofline"].get("frac_gemm_executed")
        except Exception as e:          # noqa: BLE001
            summary["trace_bench_line_error"] = str(e)
    for name, c in pmc.items():
        if timed and name == timed[-1]:
            n_mfma = c.get("SQ_INSTS_MFMA", {}).get("avg")
            if n_mfma is not None and summary.get("edge_kernel_avg_us_in_trace"):
                summary["edge_kernel_mfma_insts_per_launch"] = n_mfma
                us = summary.get("edge_kernel_steady_avg_us_in_trace") or summary["edge_kernel_avg_us_in_trace"]
                summary["edge_kernel_frac_mfma_issued"] = n_mfma * 32768.0 / (us * 1e-6) / 2.5e15
                # MFMAs per edge and the GEMM multiply-adds they amount to (three MFMAs of 32 x 32 x 16 per GEMM term and 32 edges): what
                # bench.py::edge_frame_gemm_mac() must reproduce.  (Edges: the timed steps of the traced bench line; the counter average is over
                # ALL launches of the command incl. its warm-up steps, whose edge counts differ by a few per cent.)
                if summary.get("traced_bench_line_edges_per_step") and summary.get("edge_kernel_timed_steps_avg_us_in_trace") and summary.get("edge_kernel_steady_avg_us_in_trace"):
                    # edges per launch averaged over the launches the counters average over (warm-up steps carry more edges than the timed ones): the
                    # timed steps' edge count scaled by the ratio of the average durations (the kernel's time is proportional to its edges)
                    e_est = summary["traced_bench_line_edges_per_step"] * summary["edge_kernel_steady_avg_us_in_trace"] / summary["edge_kernel_timed_steps_avg_us_in_trace"]
                    summary["edge_kernel_edges_per_launch_estimate"] = e_est
                    summary["edge_kernel_mfma_per_32_edge_tile"] = n_mfma / e_est * 32.0
                    summary["edge_kernel_gemm_mac_per_edge_from_counters"] = n_mfma / e_est * 32.0 / 3.0 * 512.0      # three MFMAs per GEMM term, 512 MAC per edge and term
    # the four ratios of the issue profile of the timed kernel (round-4 review: computed by hand until now)
    for name, c in pmc.items():
        if timed and name == timed[-1]:
            g = lambda n: c.get(n, {}).get("avg")
            if g("SQ_INSTS_VALU") and g("SQ_INSTS_MFMA"):
                summary["edge_kernel_valu_per_mfma"] = g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA")
            if g("SQ_WAVE_CYCLES"):
                wc = g("SQ_WAVE_CYCLES")          # quad-cycles, like SQ_ACTIVE_INST_* and SQ_WAIT_*; SQ_VALU_MFMA_*_CYCLES count cycles (MI355X_MICROARCH.md)
                if g("SQ_ACTIVE_INST_VALU"):
                    summary["edge_kernel_valu_active_frac_of_wave_time"] = g("SQ_ACTIVE_INST_VALU") / wc
                if g("SQ_VALU_MFMA_BUSY_CYCLES"):
                    summary["edge_kernel_mfma_busy_frac_of_wave_time"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * wc)
                if g("SQ_WAIT_INST_ANY"):
                    summary["edge_kernel_wait_inst_any_frac"] = g("SQ_WAIT_INST_ANY") / wc
                if g("SQ_WAIT_ANY"):
                    summary["edge_kernel_wait_any_frac"] = g("SQ_WAIT_ANY") / wc
            if g("SQ_VALU_MFMA_COEXEC_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
                summary["edge_kernel_coexec_frac_of_mfma_busy"] = g("SQ_VALU_MFMA_COEXEC_CYCLES") / g("SQ_VALU_MFMA_BUSY_CYCLES")
    for name, c in pmc.items():
        if timed and name == timed[-1]:
            f = c.get("FETCH_SIZE", {}).get("avg")
            w = c.get("WRITE_SIZE", {}).get("avg")
            if f is not None and w is not None:
                summary["edge_kernel_fetch_KiB_per_launch_raw"] = f
                summary["edge_kernel_write_KiB_per_launch_raw"] = w
                summary["edge_kernel_hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    # (run on the GPU box, where only gpurun_out/ travels back: DEDF_SUMMARY_DIR=gpurun_out; then copy the two files into profiles/)
    out_dir = os.path.join(ROOT, os.environ.get("DEDF_SUMMARY_DIR", "profiles"))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.txt"), "w") as fh:
        fh.write("\n".join(out_txt) + "\n")
    with open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print("\n".join(out_txt[:20]))
    print(json.dumps({k: v for k, v in summary.items() if k != "per_kernel"}, indent=1))
    for name, c in pmc.items():
        if "k_edge" in name or "k_node" in name or "k_aggregate" in name:
            print(name[:60], {k: round(v["avg"], 1) for k, v in c.items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
