"""Collects a round's evidence under gpurun_out/<round>/ into profiles/ (tracked); round = $ROUND, default r3:
  profiles/r2_<tag>_{bench.json,kernel_stats.csv,pmc_k_path.json}   per profiled workload (tools/profile_r2.sh)
  profiles/r2_calib.json                                           calibration of FETCH_SIZE and of the VALU issue peak (tools/calib_r2.sh)
  profiles/r2_pmc.json                                             what bench.py reads: per-launch counters of k_path keyed workload:mode:spp:depth
  profiles/<round>_pmc_table.md                                     the same numbers as the table DESIGN.md / BASELINE.md / profiles/README.md quote (generated, never typed)
usage: ROUND=r3 python3 tools/make_pmc_json.py tag=WORKLOAD:SPP ..."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = os.environ.get("ROUND", "r3")
SRC = os.path.join(ROOT, "gpurun_out", RND)
DST = os.path.join(ROOT, "profiles")


def calib():
    c = os.path.join(SRC, "calib")
    if not os.path.isdir(c):
        return None

    def lines(name):
        return [json.loads(ln) for ln in open(os.path.join(c, name)) if ln.startswith("{")]

    def pmc(name):
        return list(json.load(open(os.path.join(c, name))).values())

    stream, g_hbm, g_mall, valu = lines("stream.txt"), lines("gather_hbm.txt"), lines("gather_mall.txt"), lines("valu.txt")
    p_stream, p_hbm, p_mall, p_req = pmc("p_stream.json")[0], pmc("p_gather_hbm.json")[0], pmc("p_gather_mall.json")[0], pmc("p_gather_req.json")[0]
    p_valu, p_clk = pmc("p_valu.json")
    p_valu_pk, p_pk_clk = pmc("p_valu_pk.json")
    valu_pk = lines("valu_pk.txt")

    def valu_entry(rows, p, clk, what):
        ghz = clk["GRBM_GUI_ACTIVE"] / 8 / (clk["dur_ms"] * 1e6)
        return {"what": what, "wave_insts": rows[0]["wave_insts"], "SQ_INSTS_VALU": p["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU": p["SQ_ACTIVE_INST_VALU"],
                "best_ms": min(r["ms"] for r in rows), "g_wave_insts_per_s_best": max(r["g_wave_insts_per_s"] for r in rows),
                "ms_under_pmc": p["dur_ms"], "clock_ghz_under_pmc": ghz,
                "simd_cycles_per_wave_inst_under_pmc": 1024 * ghz / (p["SQ_INSTS_VALU"] / (p["dur_ms"] * 1e6))}
    best = lambda rows, k: max(r[k] for r in rows)
    out = {
        "how": "tools/calib.hip through tools/calib_r2.sh on one MI355X; FETCH_SIZE is reported in KiB",
        "stream_16B_per_lane": {"bytes_read": stream[0]["bytes_read"], "best_gb_per_s": best(stream, "gb_per_s"),
                                "FETCH_SIZE_bytes": p_stream["FETCH_SIZE"] * 1024, "fetch_size_over_bytes": p_stream["FETCH_SIZE"] * 1024 / stream[0]["bytes_read"]},
        "gather_64B_records_hbm_4GiB": {"bytes_read": g_hbm[0]["bytes_read"], "best_gb_per_s": best(g_hbm, "gb_per_s"),
                                        "FETCH_SIZE_bytes": p_hbm["FETCH_SIZE"] * 1024, "fetch_size_over_bytes": p_hbm["FETCH_SIZE"] * 1024 / g_hbm[0]["bytes_read"],
                                        "TCC_EA0_RDREQ": p_req["TCC_EA0_RDREQ_sum"], "records": g_hbm[0]["bytes_read"] // 64,
                                        "g_requests_per_s": best(g_hbm, "gb_per_s") / 64},
        "gather_64B_records_infinity_cache_128MiB": {"bytes_read": g_mall[0]["bytes_read"], "best_gb_per_s": best(g_mall[1:], "gb_per_s"),
                                                     "FETCH_SIZE_bytes": p_mall["FETCH_SIZE"] * 1024,
                                                     "fetch_size_over_bytes": p_mall["FETCH_SIZE"] * 1024 / g_mall[0]["bytes_read"],
                                                     "g_requests_per_s": best(g_mall[1:], "gb_per_s") / 64},
        "gather_64B_records_by_table_size": {str(r["table_mib"]) + "MiB": {"g_requests_per_s": r["g_records_per_s"], "gb_per_s": r["gb_per_s"]}
                                             for r in (lines("gather_sizes.txt") if os.path.exists(os.path.join(c, "gather_sizes.txt")) else [])},
        "gather_adjacent_sectors": {k: {"hbm_4GiB_g_records_per_s": best(lines(f"{k}_hbm.txt"), "g_records_per_s"), "hbm_4GiB_gb_per_s": best(lines(f"{k}_hbm.txt"), "gb_per_s"),
                                         "infinity_cache_128MiB_g_records_per_s": best(lines(f"{k}_mall.txt")[1:], "g_records_per_s"),
                                         "infinity_cache_128MiB_gb_per_s": best(lines(f"{k}_mall.txt")[1:], "gb_per_s")}
                                    for k in ("gather128", "gather256") if os.path.exists(os.path.join(c, f"{k}_hbm.txt"))},
        "valu_issue_plain": valu_entry(valu, p_valu, p_clk, "independent v_mul_f32 / v_add_f32 (inline asm), 6144 waves = 6 per SIMD"),
        "valu_issue_packed": valu_entry(valu_pk, p_valu_pk, p_pk_clk, "the same with v_pk_mul_f32 / v_pk_add_f32 (two fp32 operations per lane per instruction)"),
    }
    if os.path.exists(os.path.join(c, "coop.txt")):
        # tools/coop.hip: every wave a dependent chain of random 64-B records, 4 loads in flight, 6 waves per SIMD
        rows = lines("coop.txt")
        by = {}
        for r in rows:
            e = by.setdefault(f"{r['table_kib']}KiB", {}).setdefault(r["mode"], {"g_wave_loads_per_s": 0.0})
            e["g_wave_loads_per_s"] = max(e["g_wave_loads_per_s"], r["g_wave_loads_per_s"])
        for t in by.values():
            for m, e in t.items():
                e["cu_cycles_per_wave_load_at_2.3GHz"] = 256 * 2.3 / e["g_wave_loads_per_s"]
                if m == "lane":
                    e["g_records_per_s"] = e["g_wave_loads_per_s"] * 16  # 4 wave-loads fetch 64 records
        out["vector_memory_pipe"] = {
            "how": "tools/coop.hip: modes lane = 4 x buffer_load_dwordx4 per lane and round (k_path's node fetch), lane_16_first / lane_16_spread = the same with 16 of 64 lanes "
                   "active, lane_dwordxN = loads of N dwords, quad = the 4 lanes of a quad share 4 records (16 segments per instruction)",
            "by_table_size": by,
            "finding": "a vector load costs the CU's memory pipe 16 - 21 cycles whatever its width (dword 18, dwordx4 20.5) and however few lanes are active (16 of 64: 15.6 - 17)",
        }
    json.dump(out, open(os.path.join(DST, f"{RND}_calib.json"), "w"), indent=1)
    return out


def write_table(table):
    """profiles/<round>_pmc_table.md: per workload, what the counters and the bench line of the same profiling run say."""
    cal = table.get("_calibration", {})
    rows = ["| workload | Grays/s | k_path ms per launch | node tests / ray | prim tests / ray | algorithmic GB per launch (over HBM peak) | fabric GB per launch | HBM frac | L2 hit | L2-miss requests G/s (of the gather rate) | vector loads G wave-inst/s (of the L1-resident rate) | VALU G wave-inst/s (frac) | lanes active | lane frac | waves waiting |",
            "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for key in sorted(k for k in table if not k.startswith("_")):
        p = table[key]
        wl, _, spp, _ = key.split(":")
        tag = p["source"].split("_")[1]
        b = json.load(open(os.path.join(DST, f"{RND}_{tag}_bench.json")))
        r = b["roofline"]
        ms = r["kernel_busy_ms"] / r["launches"]
        hbm = p["hbm_bytes_per_launch"] / ms / 1e6
        req = p["ea_read_requests_per_launch"] / ms / 1e6
        peak = cal.get("fabric_gather_peak_g_requests_per_s_hbm" if wl == "C5" else "fabric_gather_peak_g_requests_per_s_infinity_cache", float("nan"))
        valu = p["valu_wave_insts_per_launch"] / ms / 1e6
        vmem, vmem_peak = p["vmem_rd_insts_per_launch"] / ms / 1e6, cal.get("vmem_peak_g_wave_loads_per_s_l1_resident_full_waves", float("nan"))
        rows.append(f"| {wl} ({spp} spp / launch) | {b['value'] / 1e3:.2f} | {ms:.1f} | {r['algorithmic']['node_tests_per_ray']:.1f} | {r['algorithmic']['prim_tests_per_ray']:.2f} | "
                    f"{r['algorithmic']['bytes_per_launch'] / 1e9:.0f} ({r['algorithmic']['over_hbm_peak']:.2f}) | {p['hbm_bytes_per_launch'] / 1e9:.1f} | {hbm / 8000:.2f} | "
                    f"{p['l2_hit_rate']:.2f} | {req:.1f} ({req / peak:.2f} of {'HBM-resident (1-GiB table)' if wl == 'C5' else 'Infinity-Cache-resident'} {peak:.1f}) | {vmem:.1f} ({vmem / vmem_peak:.2f} of {vmem_peak:.1f}) | {valu:.0f} ({valu / 1228.8:.2f}) | "
                    f"{p['lane_util']:.2f} | {valu / 1228.8 * p['lane_util']:.2f} | {p['wave_wait_any_frac']:.2f} |")
    head = (f"Generated by tools/make_pmc_json.py from profiles/{RND}_pmc.json and profiles/{RND}_<tag>_bench.json (one profiling run per workload: bench line, rocprofv3 kernel stats "
            f"and one --pmc pass per counter group of the same command).  Rates use the live kernel time of that run's bench line.\n\n")
    open(os.path.join(DST, f"{RND}_pmc_table.md"), "w").write(head + "\n".join(rows) + "\n")
    # the same table between the markers of the documents that quote it
    for doc in (os.path.join(DST, "README.md"), os.path.join(ROOT, "DESIGN.md"), os.path.join(ROOT, "BASELINE.md")):
        text = open(doc).read()
        a, b = text.find("<!-- pmc-table:begin -->"), text.find("<!-- pmc-table:end -->")
        if a >= 0 and b > a:
            text = text[:a] + "<!-- pmc-table:begin -->\n" + f"(generated from profiles/{RND}_pmc.json by tools/make_pmc_json.py)\n\n" + "\n".join(rows) + "\n" + text[b:]
            open(doc, "w").write(text)


def main():
    cal = calib()
    path = os.path.join(DST, f"{RND}_pmc.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    for arg in sys.argv[1:]:
        tag, rest = arg.split("=")
        wl, spp = rest.split(":")
        d = os.path.join(SRC, tag)
        p = json.load(open(os.path.join(d, "pmc.json")))
        shutil.copy(os.path.join(d, "pmc.json"), os.path.join(DST, f"{RND}_{tag}_pmc_k_path.json"))
        shutil.copy(os.path.join(d, "bench.json"), os.path.join(DST, f"{RND}_{tag}_bench.json"))
        if os.path.exists(os.path.join(d, "kernel_stats.csv")):
            shutil.copy(os.path.join(d, "kernel_stats.csv"), os.path.join(DST, f"{RND}_{tag}_kernel_stats.csv"))
        fetch, write = p["fetch"]["FETCH_SIZE"] * 1024, p["write"]["WRITE_SIZE"] * 1024
        sq1, sq2, tcc, grbm = p["sq1"], p["sq2"], p["tcc"], p["grbm"]
        table[f"{wl}:path:{spp}:32"] = {
            "source": f"profiles/{RND}_{tag}_pmc_k_path.json (tools/profile_round.sh {tag} {wl} {spp}: one rocprofv3 --pmc pass per counter group over `bench.py --steps 1 --warmup 1`)",
            "kernel_ms_under_pmc": sq1["dur_ms"],
            "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
            "hbm_note": "FETCH_SIZE + WRITE_SIZE (KiB x 1024) per k_path launch.  FETCH_SIZE = TCC_EA0_RDREQ x 64 B; for this kernel's read shape (one 64-B-aligned "
                        "record per lane) profiles/" + RND + "_calib.json shows it reads the bytes exactly (ratio 1.00; the x2 of MI355X_MICROARCH.md applies to wide "
                        "streaming reads, ratio 0.50 there), so no correction is applied.  These are L2-miss requests to the fabric: Infinity-Cache hits are "
                        "included, HBM itself carries at most this.",
            "ea_read_requests_per_launch": tcc["TCC_EA0_RDREQ_sum"],
            "l2_hit_rate": tcc["TCC_HIT_sum"] / (tcc["TCC_HIT_sum"] + tcc["TCC_MISS_sum"]),
            "valu_wave_insts_per_launch": sq1["SQ_INSTS_VALU"],
            "lane_util": sq2["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq1["SQ_ACTIVE_INST_VALU"]),
            "salu_insts_per_launch": sq2["SQ_INSTS_SALU"], "vmem_rd_insts_per_launch": sq2["SQ_INSTS_VMEM_RD"],
            "wave_wait_any_frac": sq1["SQ_WAIT_ANY"] / sq1["SQ_WAVE_CYCLES"],
            "clock_ghz_under_pmc": grbm["GRBM_GUI_ACTIVE"] / 8 / (grbm["dur_ms"] * 1e6),
        }
    if cal:
        ic = cal["gather_64B_records_infinity_cache_128MiB"]["g_requests_per_s"]
        # HBM-resident working sets here are 1 - 2 GiB (C5: 1.4 GB): the rate of a 1-GiB table, not of the 4-GiB one behind the cliff (19 G/s) that round 2 used
        hbm = cal["gather_64B_records_by_table_size"].get("1024MiB", cal["gather_64B_records_hbm_4GiB"])["g_requests_per_s"]
        c = {"fabric_gather_read_once_g_requests_per_s_infinity_cache": ic, "fabric_gather_read_once_g_requests_per_s_hbm": hbm,
             "fabric_gather_peak_g_requests_per_s_hbm_tables_of_4GiB_and_more": cal["gather_64B_records_hbm_4GiB"]["g_requests_per_s"]}
        vm = cal.get("vector_memory_pipe", {}).get("by_table_size")
        if vm:
            # dependent chains with 4 loads in flight per wave (tools/coop.hip) gather 2 - 3 times faster than the read-once kernel of tools/calib.hip: the higher rate is the roof
            ic, hbm = max(ic, vm["65536KiB"]["lane"]["g_records_per_s"]), max(hbm, vm["1048576KiB"]["lane"]["g_records_per_s"])
            c["vmem_peak_g_wave_loads_per_s_l1_resident_full_waves"] = vm["16KiB"]["lane"]["g_wave_loads_per_s"]
            c["vmem_peak_g_wave_loads_per_s_l1_resident_16_lanes"] = vm["16KiB"]["lane_16_spread"]["g_wave_loads_per_s"]
            c["vmem_peak_g_wave_loads_per_s_l2_resident_full_waves"] = vm["2048KiB"]["lane"]["g_wave_loads_per_s"]
            c["vmem_peak_g_wave_loads_per_s_l2_resident_16_lanes"] = vm["2048KiB"]["lane_16_spread"]["g_wave_loads_per_s"]
        c.update({"fabric_gather_peak_g_requests_per_s_infinity_cache": ic, "fabric_gather_peak_g_requests_per_s_hbm": hbm,
                  "valu_plain_measured_g_wave_insts_per_s": cal["valu_issue_plain"]["g_wave_insts_per_s_best"],
                  "valu_plain_simd_cycles_per_wave_inst": cal["valu_issue_plain"]["simd_cycles_per_wave_inst_under_pmc"],
                  "valu_packed_simd_cycles_per_wave_inst": cal["valu_issue_packed"]["simd_cycles_per_wave_inst_under_pmc"],
                  "source": f"profiles/{RND}_calib.json"})
        table["_calibration"] = c
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    write_table(table)
    print(json.dumps(table, indent=1)[:3000])


if __name__ == "__main__":
    main()
