"""Turn gpurun_out/prof_<ROUND><TAG>/* (tools/profile_round.sh) into the committed summaries under profiles/:
  <ROUND><TAG>_kernel_stats_{serial,default}.{csv,md}, <ROUND><TAG>_pmc_counters.md, <ROUND><TAG>_conv_traffic.json
usage: ROUND=r03 TAG=_cfg4_1280x384 python tools/make_profile_md.py [frames_in_stats_run=217] [frames_in_pmc_run=8]"""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT = os.environ.get("ROUND", "r03") + os.environ.get("TAG", "")
SRC, DST = os.path.join(ROOT, "gpurun_out", "prof_" + RT), os.path.join(ROOT, "profiles")
frames = float(sys.argv[1]) if len(sys.argv) > 1 else 217.0


def bench_line(mode):
    for line in open(os.path.join(SRC, f"bench_{mode}.log")):
        if line.startswith('{"metric'):
            return json.loads(line)
    return {}


stats = {}
for mode in ("serial", "default"):
    rows = list(csv.DictReader(open(os.path.join(SRC, f"{mode}_kernel_stats.csv"))))
    shutil.copy(os.path.join(SRC, f"{mode}_kernel_stats.csv"), os.path.join(DST, f"{RT}_kernel_stats_{mode}.csv"))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    conv = [r for r in rows if "conv_bf16_kernel" in r["Name"]]
    conv32 = [r for r in rows if "conv_mfma_kernel" in r["Name"] or "conv_quad_kernel" in r["Name"]]
    spl = [r for r in rows if "split_bf16_kernel" in r["Name"]]
    cn, ct = sum(int(r["Calls"]) for r in conv), sum(int(r["TotalDurationNs"]) for r in conv)
    b = bench_line(mode)
    stats[mode] = dict(rows=rows, tot=tot, conv_calls=cn, conv_ns=ct)
    with open(os.path.join(DST, f"{RT}_kernel_stats_{mode}.md"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats: {RT}, {'serial streams' if mode == 'serial' else 'default schedule (side streams ON)'}\n\n")
        f.write(f"Command (tools/profile_round.sh): `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --tune-db <db>{' --serial-streams' if mode == 'serial' else ''}` "
                f"({frames:.0f} frames: 1 priming + 150 pre-warm + 5 warm-up + 60 timed + 1 eager roofline frame; launch configurations = the shipped codd_amd/tuned/mi355x.json, so no tuning launches are in the statistics).\n\n")
        f.write(f"bench.py under the profiler: {b.get('value')} frames/s, {b.get('ms_per_step')} ms/step (the profiler slows the run and inflates bench.py's own event brackets: "
                f"conv_ms_per_frame {(b.get('roofline') or {}).get('conv_ms_per_frame')} here; see bench_plain.log for the un-profiled run).\n\n")
        f.write(f"Total kernel time {tot/1e6:.1f} ms over {frames:.0f} frames = {tot/frames/1e6:.2f} ms/frame (sum of kernel durations"
                f"{'; with side streams kernels overlap, so this exceeds the wall time' if mode == 'default' else ''}).\n\n")
        f.write(f"conv_bf16_kernel<*> family (split-bf16 convolutions): {cn} launches = {cn/frames:.0f}/frame, {ct/frames/1e6:.2f} ms/frame, avg {ct/cn/1e3:.1f} us/launch; "
                f"their re-layout passes split_bf16_kernel: {sum(int(r['Calls']) for r in spl)/frames:.0f}/frame, {sum(int(r['TotalDurationNs']) for r in spl)/frames/1e6:.2f} ms/frame; "
                f"exact-fp32 family conv_mfma_kernel<*> + conv_quad_kernel<*> (HITNet, all-pairs, layers the tuner keeps on fp32): "
                f"{sum(int(r['Calls']) for r in conv32)/frames:.0f}/frame, {sum(int(r['TotalDurationNs']) for r in conv32)/frames/1e6:.2f} ms/frame.\n\n")
        f.write("| kernel | calls | ms/frame | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:36]:
            f.write(f"| `{r['Name'][:80]}` | {r['Calls']} | {int(r['TotalDurationNs'])/frames/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |\n")

# ---- PMC ---------------------------------------------------------------------------------------------------------
def load_pmc(tag):
    d = {}
    for r in csv.DictReader(open(os.path.join(SRC, f"pmc_{tag}.csv"))):
        d[(r["kernel"], r["counter"])] = (float(r["sum"]), int(r["launches"]))
    return d


fetch, write, mf = load_pmc("FETCH_SIZE"), load_pmc("WRITE_SIZE"), load_pmc("SQ_VALU_MFMA_BUSY_CYCLES")
serial_avg = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in stats["serial"]["rows"]}
kernels = sorted({k for k, _ in fetch})
conv_f = sum(v[0] for (k, c), v in fetch.items() if "conv_bf16_kernel" in k)
conv_w = sum(v[0] for (k, c), v in write.items() if "conv_bf16_kernel" in k)
conv_n = sum(v[1] for (k, c), v in fetch.items() if "conv_bf16_kernel" in k)
# FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts
# wide coalesced reads at half their size -> traffic upper estimate = 2 * FETCH + WRITE
traffic_per_launch = (2 * conv_f + conv_w) * 1024.0 / conv_n
json.dump(dict(family="bf16" if "bf16" in RT else "split_bf16", kernel="conv_bf16_kernel<*>", launches=conv_n, fetch_kib_per_launch=conv_f / conv_n,
               write_kib_per_launch=conv_w / conv_n, traffic_bytes_per_launch=traffic_per_launch,
               correction="traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE counts wide reads at half size)",
               source="tools/profile_round.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) on "
                      "bench.py --serial-streams --no-graph --steps 4"),
          open(os.path.join(DST, RT + "_conv_traffic.json"), "w"), indent=1)
with open(os.path.join(DST, RT + "_pmc_counters.md"), "w") as f:
    f.write(f"# rocprofv3 PMC counters per kernel ({RT})\n\n")
    f.write("Collected by `tools/profile_round.sh`: three separate passes of `rocprofv3 --kernel-trace --pmc <set> -- python bench.py "
            "--no-cpu-baseline --serial-streams --steps 4 --prewarm 2 --warmup 1 --no-graph` with the sets `FETCH_SIZE`, `WRITE_SIZE`, "
            "`SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES`.  FETCH_SIZE / WRITE_SIZE are KiB; per MI355X_MICROARCH.md "
            "(HBM section) FETCH_SIZE on gfx950 counts wide coalesced reads at half their size, so `traffic = (2*FETCH + WRITE) KiB` "
            "(an upper estimate for narrow gathers).  Durations are the un-counted serial-stream averages of "
            f"`{RT}_kernel_stats_serial.csv`.\n\n")
    f.write(f"## Convolution families\n\nAll `conv_bf16_kernel<*>` (split-bf16) launches: FETCH {conv_f/conv_n:.0f} KiB + WRITE {conv_w/conv_n:.0f} KiB per launch "
            f"-> L2-miss-side traffic {(traffic_per_launch)/1e6:.2f} MB per launch ({conv_n} launches in the pass).\n\n")
    # which layer(s) an instantiation of conv_bf16_kernel<PGW, CGW, A, B, TERMS, OUTF, KS> serves: bench.py's roofline.instantiations
    inst_layers = ((bench_line("serial").get("roofline") or {}).get("instantiations")) or {}

    def layers_of(k):
        for inst, ls in inst_layers.items():
            if "conv_bf16_kernel" + inst in k.replace("void ", ""):
                return "; ".join(f"{l['layer']} x{l['launches_per_frame']} ({l['workgroups']} workgroups, {l['us_per_launch']} us by HIP events)" for l in ls)
        return ""
    f.write("| instantiation <NW,NPB,MB,WREG,IREG|QREG> / <PGW,CGW,A,B,TERMS,OUTF,KS> | launches in pass | MFMA busy cycles/launch | MFMA pipe utilisation | FETCH KiB/launch | WRITE KiB/launch | layers of one frame (split-bf16 family) |\n|---|---|---|---|---|---|---|\n")
    for k in kernels:
        if "conv_mfma" not in k and "conv_quad" not in k and "conv_bf16" not in k:
            continue
        busy, n = mf.get((k, "SQ_VALU_MFMA_BUSY_CYCLES"), (0, 1))
        act, _ = mf.get((k, "GRBM_GUI_ACTIVE"), (0, 1))
        util = busy / (act / 8.0 * 1024.0) if act else float("nan")
        f.write(f"| `{k.replace('void ', '').replace('(ConvK)', '').replace('(ConvB)', '')}` | {n} | {busy/n:.3g} | {util:.3f} | {fetch[(k,'FETCH_SIZE')][0]/fetch[(k,'FETCH_SIZE')][1]:.0f} | "
                f"{write.get((k,'WRITE_SIZE'),(0,1))[0]/write.get((k,'WRITE_SIZE'),(0,1))[1]:.0f} | {layers_of(k)} |\n")
    f.write("\n`util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCD x 1024 SIMD)`.\n\n## Other kernels (HBM / latency bound)\n\n")
    f.write("| kernel | launches in pass | avg us (serial) | FETCH KiB/launch | WRITE KiB/launch | traffic MB/launch (2F+W) | GB/s at the serial duration |\n|---|---|---|---|---|---|---|\n")
    for k in kernels:
        if "conv_mfma" in k or "conv_quad" in k or "conv_bf16" in k or "rocclr" in k or "at::native" in k:  # torch kernels: eager set-up only
            continue
        fs, n = fetch[(k, "FETCH_SIZE")]
        ws, _ = write.get((k, "WRITE_SIZE"), (0, 1))
        tr = (2 * fs + ws) * 1024.0 / n
        us = serial_avg.get(k)
        f.write(f"| `{k[:60]}` | {n} | {us:.1f} | {fs/n:.0f} | {ws/n:.0f} | {tr/1e6:.2f} | {tr/us/1e3:.0f} |\n" if us else
                f"| `{k[:60]}` | {n} | - | {fs/n:.0f} | {ws/n:.0f} | {tr/1e6:.2f} | - |\n")
print("conv traffic per launch: %.2f MB" % (traffic_per_launch / 1e6))
