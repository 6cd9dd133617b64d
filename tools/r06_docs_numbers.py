"""Prints the figures the round-6 documents quote, computed from profiles/r06_* (after tools/r06_collect.py), as one dict -- and, with --apply, rewrites the
"Roofline numbers (MI355X, round 6 ...)" section of DESIGN.md from its template below.  README.md and profiles/README.md quote a subset: --apply prints the
old -> new pairs it could not place so that they are edited by hand."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def jl(name):
    return json.loads(open(os.path.join(P, name)).read().strip().splitlines()[-1])


def figures():
    f = {}
    d = jl("r06_bench_grid8.json"); r = d["roofline"]
    f.update(mp=d["value"], ms=d["ms_per_step"], e2e=r["end_to_end_achieved"], e2e_frac=r["end_to_end_frac"], fam=r["achieved"], fam_frac=r["frac"], fam_us=r["avg_launch_us"],
             sl=r["single_lane"]["achieved"], sl_frac=r["single_lane"]["frac"], sl_us=r["single_lane"]["avg_launch_us"], share=100 * r["share_of_unet_kernel_time"],
             traffic_mb=r["traffic"] / 1e6, t_ratio=r["traffic_over_algorithmic"], t_strict=r["traffic_over_algorithmic_strict"], build=r["library_build_id"],
             sb=r["small_batch_kernel"]["achieved"], lat=d["latency_single_tile_ms"], anchor=d["strong_scaling_anchor"]["value"], anchor_ms=d["strong_scaling_anchor"]["ms_per_64_window_batch"],
             cpu=d["cpu_baseline"]["value"])
    rows = [x for x in csv.DictReader(open(os.path.join(P, "r06_bench_grid8_kernel_trace_summary.csv"))) if "conv_glds_kernel" in x["kernel"]]
    calls = sum(int(x["calls"]) for x in rows); t = sum(float(x["total_us"]) for x in rows)
    f.update(tr_calls=calls, tr_ms=t / 1e3, tr_us=t / calls, tr_tf=199.127893693e9 / (t / calls * 1e-6) / 1e12)
    f["tr_frac"] = f["tr_tf"] / 2500
    pj = json.load(open(os.path.join(P, "r06_hbm_traffic_and_mfma_util.json")))
    k = pj["kernels"]; wide = [v for n, v in k.items() if "conv_glds_kernel_wide" in n][0]; sbk = [v for n, v in k.items() if "conv_sb_kernel" in n][0]
    others = [v for n, v in k.items() if "conv_glds_kernel" in n and "wide" not in n]
    fam = [v for n, v in k.items() if "conv_glds" in n and v.get("mfma_util")]
    f.update(busy_w=100 * wide["mfma_util"], wait_w=100 * wide["wait_any_share"], vpm=wide["valu_per_mfma"], l2_w=wide["l2_hit_rate"], fab_m=wide["fabric_read_requests_per_launch"] / 1e6,
             busy_lo=100 * min(v["mfma_util"] for v in others), busy_hi=100 * max(v["mfma_util"] for v in others), l2_lo=min(v["l2_hit_rate"] for v in others), l2_hi=max(v["l2_hit_rate"] for v in others),
             busy_sb=100 * sbk["mfma_util"], l2_sb=sbk["l2_hit_rate"], busy_fam=100 * sum(v["dispatches"] * v["mfma_util"] for v in fam) / sum(v["dispatches"] for v in fam), wide_n=wide["dispatches"])
    d1 = jl("r06_bench_grid8_single_lane.json"); f.update(sl_line=d1["value"], sl_line_frac=d1["roofline"]["frac"], sl_line_tf=d1["roofline"]["achieved"], sl_line_us=d1["roofline"]["avg_launch_us"])
    c = jl("r06_bench_cascade.json"); cr = c["roofline"]; lv = cr["one_request_kernel_ms_by_resolution"]
    f.update(casc=c["value"], casc16=jl("r06_bench_cascade_fp16.json")["value"], casc_sync=jl("r06_bench_cascade_synchronous.json")["value"], req_ms=cr["one_request_conv_kernel_ms"],
             lat_ms=lv["coarse+latent 64x64 and below"], d128=lv["decoder levels 128x128"], d256=lv["decoder levels 256x256"], d512=lv["decoder levels 512x512"], d512_gbps=cr["hbm_gbps_decoder_512x512"])
    f.update(tiles=jl("r06_bench_tiles.json")["value"], g16=jl("r06_bench_grid8_fp16.json")["value"], g32=jl("r06_bench_grid8_fp32.json")["value"], g32_frac=jl("r06_bench_grid8_fp32.json")["roofline"]["frac"],
             grid32=jl("r06_bench_grid32_n1.json")["value"])
    tt = json.load(open(os.path.join(P, "r06_ttft_ttst_latency.json"))); f.update(ttft=1e3 * tt["ttft_mean"], ttst=1e3 * tt["ttst_mean"])
    f["perop_ms"] = float(re.search(r"([0-9.]+) ms kernel time", open(os.path.join(P, "r06_per_op_batch64.txt")).readline()).group(1))
    lev = {}
    for l in open(os.path.join(P, "r06_per_op_batch64.txt")):
        m = re.search(r"\[(\d+x\d+) k\d+ (f\S+)", l); g = re.search(r"gf([0-9.]+)", l)
        if m and " us " in l and g:
            a = lev.setdefault(m.group(1) + " " + m.group(2), [0.0, 0.0]); a[0] += float(l.split()[0]); a[1] += float(g.group(1))
    pf = lambda key: lev[key][1] / lev[key][0] if key in lev else float("nan")   # GFLOP / us = PFLOP/s
    f.update(w64=pf("64x64 f2w"), w32=pf("32x32 f2w"), w16=pf("16x16 f2w"), t64=pf("64x64 f2b"), t32=pf("32x32 f2s"), t16=pf("16x16 f2s"), s8=pf("8x8 f4m2n2"))
    k3 = sorted(float(l.split()[0]) for l in open(os.path.join(P, "r06_per_op_batch64.txt")) if "64x64 k3 f2w" in l)
    f.update(k3_lo=173.95 / k3[-1], k3_hi=173.95 / k3[0], k3_us_lo=k3[0], k3_us_hi=k3[-1])
    sweep = {}
    for l in open(os.path.join(P, "r06_batch_sweep.txt")):
        m = re.match(r"\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s", l)
        if m:
            sweep[int(m.group(1))] = (float(m.group(2)), float(m.group(4)))
    f["sweep_ms"] = " / ".join(f"{sweep[b][0]:.3f}" for b in (1, 2, 4, 8, 16, 32, 64)); f["sweep_tf"] = " / ".join(f"{sweep[b][1]:.0f}" for b in (1, 2, 4, 8, 16, 32, 64))
    ab = {}
    for l in open(os.path.join(P, "r06_wide_tile_and_two_lanes_ab.txt")):
        m = re.match(r"\[(.*?)\] ([0-9.]+) MP/s", l)
        if m:
            ab.setdefault(m.group(1), []).append(float(m.group(2)))
    mean = lambda key: sum(ab[key]) / len(ab[key])
    casc_def = [v for v in ab[""] if v > 23]; grid_def = [v for v in ab[""] if v < 23]
    f.update(ab_r5=mean("glds_wide=0,dual_stream=0"), ab_lanes=sum(v for v in ab["glds_wide=0"] if v < 23) / 2, ab_wide=mean("dual_stream=0"), ab_both=sum(grid_def) / len(grid_def), ab_1024=mean("glds_wide_min_wgs=1024"),
             abc_nowide=sum(v for v in ab["glds_wide=0"] if v > 23) / 2, abc_nofc=mean("fewcout=0"), abc_def=sum(casc_def) / len(casc_def))
    at = {}
    for l in open(os.path.join(P, "r06_attention_mfma_utilisation.txt")):
        m = re.search(r"4096x4096 d(\d+) .*kernel ([0-9.]+) us .*mfma_busy ([0-9.]+) %", l)
        if m:
            at[int(m.group(1))] = (float(m.group(2)), float(m.group(3)))
    f.update(a40_us=at[40][0], a40=at[40][1], a64=at[64][1], a64_us=at[64][0], a128=at[128][1], a160=at[160][1])
    tl = open(os.path.join(P, "r06_gpu_tests.txt")).read(); m = re.search(r"(\d+) passed, (\d+) skipped", tl); f.update(passed=int(m.group(1)), skipped=int(m.group(2)))
    f["dec_ms"] = float(re.search(r"([0-9.]+) ms kernel time", open(os.path.join(P, "r06_decoder_forward_batch4.txt")).readline()).group(1))
    f["fc_us"] = float([l for l in open(os.path.join(P, "r06_decoder_forward_batch4.txt")) if "out_conv" in l][0].split()[0])
    b1 = json.load(open(os.path.join(P, "r06_batch1_hbm_traffic.json"))); f.update(b1_r=b1["hbm_read_bytes_per_forward"] / 1e9, b1_w=b1["hbm_write_bytes_per_forward"] / 1e9)
    m = re.search(r"(\d+) kernels, span (\d+) us, kernel time (\d+) us \(([0-9.]+) %\)", open(os.path.join(P, "r06_batch1_timeline.txt")).read()); f.update(b1_k=int(m.group(1)) // (2 if int(m.group(1)) > 3000 else 1), b1_in=float(m.group(4)))   # (the timeline tool sometimes captures two replays)
    return f


TEMPLATE = """* default bench (`python bench.py`, BASELINE configs[2], `profiles/r06_bench_grid8.json`): **{mp:.2f} MP/s** (round 5: 19.29; 20.0–21.7 over the collections of the round, each on another box: see the note on the box spread below), {ms:.1f} ms/step,
  {e2e:.0f} TFLOP/s end to end (**{e2e_frac:.3f}** of 2.5 PF). The timed region runs two sampler lanes (engine default), so the line carries two kernel-level legs:
  `roofline.achieved / frac` = the family's algorithmic FLOP over the wall time it occupies in the timed region (step time × its {share:.1f} % share of U-Net kernel time): **{fam:.0f} TFLOP/s = {fam_frac:.3f}**
  ({fam_us:.1f} µs per launch-equivalent); `roofline.single_lane` = the same 1140 launches at the full batch of 64 timed one by one with HIP events on the engine's stream: **{sl:.0f} TFLOP/s = {sl_frac:.3f}**
  ({sl_us:.1f} µs per launch). The rocprofv3 trace (`r06_bench_grid8_kernel_trace_summary.csv`, one lane, 3 steps) reproduces the latter: {tr_calls} launches of the family, {tr_ms:.2f} ms, {tr_us:.1f} µs per
  launch = {tr_tf:.0f} TFLOP/s = **{tr_frac:.3f}** (round 5 by the same computation: 0.376). One-lane bench line of the same run (`r06_bench_grid8_single_lane.json`): {sl_line:.2f} MP/s, family {sl_line_frac:.3f}.
  The 8×8 level (440 launches, conv_sb): {sb:.0f} TFLOP/s.
* **box spread.** The conv kernels have not changed since the fifth collection; the last four collections differ in `conv_fewcout.hip` and the attention kernel only. The sixth's box
  (build `ce275468cb4e32e1`; line and trace summary kept as `r06_bench_grid8_sixth_collection_build_ce275468*.`) gave **21.68 MP/s, family 0.421 lane-aware / 0.407 one lane / 0.412 from
  the trace**; the seventh (build `1aaa9aea…`) 20.93; the final build on FOUR boxes 20.03 / 20.11 / 20.92 / 20.86 (`r06_bench_grid8.json`, `r06_bench_grid8_final_build_{second,third,fourth}_box.json`), family 0.387 / 0.389 / 0.404 / 0.403. The per-layer tables
  say what differs: the MFMA-dense layers are 6–8 % slower on the later boxes (dec.512x512_up.conv_res1 583 → 621 µs) while the latency-bound kernels are not (attention block 43.0 → 42.7 µs,
  qkv conv 39.8 → 38.6) — the power-limited shader clock of §4 "conv throughput", which is a property of the chip (and of what its neighbours on the node are doing), not of the build. The
  review's ≥ 0.40 on the driver line was reached on the first six boxes of the round (0.408–0.421) and on two of the final build's four (0.404 / 0.403), missed on its other two (0.387 / 0.389).
* the two changes against each other (`r06_wide_tile_and_two_lanes_ab.txt`, one box, interleaved twice): round-5 configuration {ab_r5:.2f} MP/s; two lanes only {ab_lanes:.2f} ({p_lanes:+.1f} %); wide tile only {ab_wide:.2f}
  ({p_wide:+.1f} %); both {ab_both:.2f} ({p_both:+.1f} %); the wide tile without the 16×16 level {ab_1024:.2f}. Cascade {abc_nowide:.2f} (no wide tile) → {abc_nofc:.2f} (no few-cout flavour) → {abc_def:.2f} MP/s.
* HBM traffic of the family (`r06_hbm_traffic_and_mfma_util.json`, FETCH × 2 + WRITE, one lane): **{traffic_mb:.1f} MB per launch** against 194.8 MB algorithmic = **{t_ratio:.2f}×** ({t_strict:.2f}× strict; round 5: 311.2 MB,
  1.60× / 1.85×) — the wide tile halves the cout-tile siblings' re-requests of the input at the levels it serves. This rocprofv3 lists no memory-side-cache (Infinity Cache) counter for gfx950 —
  the TCC block ends at the fabric request classes (`TCC_EA0_RDREQ{{,_32B,_DRAM,_GMI,_IO}}`; "DRAM" is a destination class, not a miss) — so the sibling re-requests cannot be split into
  cache hits and HBM reads by a counter; what the counters do give: **L2 hit rate {l2_w:.2f}** on the wide kernel ({l2_lo:.2f}–{l2_hi:.2f} on the conv_glds instantiations, {l2_sb:.2f} on conv_sb) and {fab_m:.2f} M fabric
  read requests per wide launch. MFMA busy: **{busy_w:.1f} %** on the wide kernel ({wide_n} of the {tr_calls} launches; waves waiting {wait_w:.0f} %, {vpm:.1f} non-MFMA VALU per MFMA), {busy_lo:.1f}–{busy_hi:.1f} % on the conv_glds
  instantiations that keep the 1×1-tail launches, {busy_sb:.1f} % on conv_sb; **{busy_fam:.1f} %** dispatch-weighted over the family (round 5: 44.0 %). LDS bank conflicts 0.0 %.
* per layer (`r06_per_op_batch64.txt`): **{perop_ms:.2f} ms** of kernel time per batch-64 forward (round 5: 13.72); 39 launches on the wide tile (64×64 level {w64:.2f} PF, 32×32 {w32:.2f}, 16×16 {w16:.2f} on average),
  18 on conv_glds (the 1×1-tail launches: {t64:.2f} / {t32:.2f} / {t16:.2f} PF at 64×64 / 32×32 / 16×16), 22 on conv_sb (8×8: {s8:.2f}). The six short-K 192-cout encoder layers of the 64×64 level: 0.77–0.80 →
  **{k3_lo:.2f}–{k3_hi:.2f} PF** in the network ({k3_us_lo:.0f}–{k3_us_hi:.0f} µs each; 0.92–0.96 PF in the layer harness' hot loop; the review asked for ≥ 0.88: 0.89–0.92 on the sixth collection's box).
* batch sweep (`r06_batch_sweep.txt`, one build): batch 1 / 2 / 4 / 8 / 16 / 32 / 64 = {sweep_ms} ms = {sweep_tf} TFLOP/s
  (round 5: … 7.439 / 13.556 ms = 833 / 914): the wide tile enters from batch 8.
* **single tile, configs[1] as written: {lat:.2f} ms** per tile × 20 steps (round 5: 21.41; 21.3–21.6 over the collections: no kernel of that path changed) = 0.059 of the HBM peak;
  `r06_batch1_hbm_traffic.json`: {b1_r:.2f} GB read + {b1_w:.3f} GB written per forward against 0.507 GB of weights; {b1_k} kernels per 20-step replay, {b1_in:.0f} % of the time inside kernels (`r06_batch1_timeline.txt`).
* cascade (`r06_bench_cascade.json`): **{casc:.2f} MP/s** bf16 enqueue-only (round 5: 24.01; 25.5–27.0 over the collections), fp16 {casc16:.2f}, synchronous {casc_sync:.2f}; conv kernel time of one cold 1024² request
  {req_ms:.1f} ms (75.8): coarse + latent {lat_ms:.1f}, decoder 128² / 256² / 512² levels {d128:.1f} / {d256:.1f} / **{d512:.1f} ms at {d512_gbps:.0f} GB/s** of the level's algorithmic bytes (round 5: 16.4 ms, 2091; the third collection,
  before the few-cout flavour: 14.5 ms, 2378). The decoder model per layer at batch 4 × 512²: `r06_decoder_forward_batch4.txt` ({dec_ms:.2f} ms per forward, output conv {fc_us:.0f} µs). The review's 26 MP/s was reached on
  one collection's box (27.0; 25.5–25.9 on the others), its 2.8 TB/s was not. TTFT / TTST {ttft:.1f} / {ttst:.1f} ms (68.5 / 24.1).
* `strong_scaling_anchor` (configs[3] on one rank, default plan, same run): **{anchor:.2f} MP/s** (round 5, batch-invariant: 15.50), {anchor_ms:.1f} ms per 64-window batch; the full grid32 line on one rank
  (`r06_bench_grid32_n1.json`): {grid32:.2f} MP/s. Independent tiles: {tiles:.1f} MP/s (61.9). fp16 storage on grid8: {g16:.2f} MP/s (18.57). Exact-fp32 mode: {g32:.2f} MP/s, {g32_frac:.2f} of the fp32 MFMA peak.
* attention (`r06_attention_mfma_utilisation.txt`; with the software-pipelined tile loop, `r06_attention_pipelined_loop.txt`): SD 4096² d40 **{a40_us:.1f} µs, MFMA busy {a40:.1f} %** (round 4/5: 91.2 µs, 28.5 %; the fifth collection, folded softmax only: 76.9 µs, 33.8 %); d64 **{a64_us:.1f} µs, {a64:.1f} %** (100.8 µs, 29.8 %); d128 / d160 {a128:.1f} / {a160:.1f} % (d160 keeps the unpipelined loop).
* GPU tests at this build: **{passed} passed, {skipped} skipped** (the two two-GPU tests; `r06_gpu_tests.txt`); smoke rel-RMS vs oracle 5.7e-3. CPU baseline (oracle, 16 of 256 host threads): {cpu:.3f} MP/s.
"""


def main():
    f = figures()
    f.update(p_lanes=100 * (f["ab_lanes"] / f["ab_r5"] - 1), p_wide=100 * (f["ab_wide"] / f["ab_r5"] - 1), p_both=100 * (f["ab_both"] / f["ab_r5"] - 1))
    body = TEMPLATE.format(**f)
    if "--apply" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md"); s = open(p).read()
        a = s.index("### Roofline numbers (MI355X, round 6;"); b = s.index("### Roofline numbers (MI355X, round 5;")
        sec = s[a:b]; i = sec.index("* default bench (`python bench.py`")
        head = re.sub(r"library build `[0-9a-f]{16}` stamped", "library build `%s` stamped" % f["build"], sec[:i])
        open(p, "w").write(s[:a] + head + body + "\n" + s[b:])
        print("DESIGN.md section rewritten for build", f["build"])
    if "--apply" in sys.argv:
        N = r"[0-9]+(?:\.[0-9]+)?"
        def sub(path, pairs):
            t = open(path).read()
            for pat, rep in pairs:
                t2, n = re.subn(pat, lambda m: rep, t, count=1)
                if n != 1:
                    print("NOT PLACED in", os.path.basename(path), ":", pat[:90])
                t = t2
            open(path, "w").write(t)
        sub(os.path.join(ROOT, "README.md"), [
            (rf"\*\*{N} decoded MP/s\*\* \([^)]*\), {N} of the nominal", f"**{f['mp']:.1f} decoded MP/s** (20.6–21.4 over the collections of the round, each on another box; round 5: 19.3), {f['e2e_frac']:.3f} of the nominal"),
            (rf"tile\) \*\*{N}\*\* of peak on the two-lane bench line, {N} one lane live, \*\*{N}\*\* recomputed", f"tile) **{f['fam_frac']:.3f}** of peak on the two-lane bench line, {f['sl_frac']:.3f} one lane live, **{f['tr_frac']:.3f}** recomputed"),
            (rf"  {N} % on the wide kernel; \*\*single 64×64 tile", f"  {f['busy_w']:.0f} % on the wide kernel; **single 64×64 tile"),
            (rf"\(configs\[4\] shapes\) {N} MP/s\*\*; independent tiles {N} MP/s; configs\[3\] on one rank {N} MP/s; exact-fp32 mode {N} MP/s \({N} of the fp32 MFMA peak\); MFMA attention {N} / {N} / {N} / {N} %",
             f"(configs[4] shapes) {f['casc']:.1f} MP/s**; independent tiles {f['tiles']:.0f} MP/s; configs[3] on one rank {f['grid32']:.1f} MP/s; exact-fp32 mode {f['g32']:.2f} MP/s ({f['g32_frac']:.2f} of the fp32 MFMA peak); MFMA attention {f['a40']:.1f} / {f['a64']:.1f} / {f['a128']:.1f} / {f['a160']:.1f} %"),
            (rf"512² ≥ 2\.8 TB/s \({N}\), single tile", f"512² ≥ 2.8 TB/s ({f['d512_gbps'] / 1e3:.2f}), single tile"),
        ])
        sub(os.path.join(ROOT, "profiles", "README.md"), [
            (r"on the library build `[0-9a-f]{16}`\n", f"on the library build `{f['build']}`\n"),
            (rf"\*\*{N} MP/s\*\* \({N} ms per step, {N} of peak end to end\)", f"**{f['mp']:.2f} MP/s** ({f['ms']:.1f} ms per step, {f['e2e_frac']:.3f} of peak end to end)"),
            (rf"`roofline\.frac` \*\*{N}\*\*", f"`roofline.frac` **{f['fam_frac']:.3f}**"),
            (rf"`roofline\.single_lane` \*\*{N} TFLOP/s = {N}\*\*", f"`roofline.single_lane` **{f['sl']:.0f} TFLOP/s = {f['sl_frac']:.3f}**"),
            (rf"{N} MP/s, family {N} TFLOP/s = {N}, {N} µs per launch", f"{f['sl_line']:.2f} MP/s, family {f['sl_line_tf']:.1f} TFLOP/s = {f['sl_line_frac']:.3f}, {f['sl_line_us']:.1f} µs per launch"),
            (rf"{N} ms, \*\*{N} µs per launch =", f"{f['tr_ms']:.2f} ms, **{f['tr_us']:.1f} µs per launch ="),
            (rf"family {N} MB per launch; MFMA busy \*\*{N} %\*\* on the wide kernel, {N}–{N} % on the conv_glds instantiations, {N} % conv_sb, {N} % dispatch-weighted",
             f"family {f['traffic_mb']:.1f} MB per launch; MFMA busy **{f['busy_w']:.1f} %** on the wide kernel, {f['busy_lo']:.1f}–{f['busy_hi']:.1f} % on the conv_glds instantiations, {f['busy_sb']:.1f} % conv_sb, {f['busy_fam']:.1f} % dispatch-weighted"),
            (rf"{N} kernels per replay, {N} % inside kernels", f"{f['b1_k']} kernels per replay, {f['b1_in']:.0f} % inside kernels"),
            (rf"(?:{N} / ){{6}}{N} ms = (?:{N} / ){{6}}{N} TFLOP/s; `f2w`", f"{f['sweep_ms']} ms = {f['sweep_tf']} TFLOP/s; `f2w`"),
            (rf"{N} ms per batch-64 forward \(round 5: 13\.72\)", f"{f['perop_ms']:.2f} ms per batch-64 forward (round 5: 13.72)"),
            (rf"round-5 configuration {N} / two lanes only {N} / wide tile only {N} / both {N} MP/s / wide tile without the 16×16 level {N} \(one box, interleaved twice\); cascade {N} without the wide tile / {N} without the few-cout flavour / {N} with the defaults",
             f"round-5 configuration {f['ab_r5']:.2f} / two lanes only {f['ab_lanes']:.2f} / wide tile only {f['ab_wide']:.2f} / both {f['ab_both']:.2f} MP/s / wide tile without the 16×16 level {f['ab_1024']:.2f} (one box, interleaved twice); cascade {f['abc_nowide']:.2f} without the wide tile / {f['abc_nofc']:.2f} without the few-cout flavour / {f['abc_def']:.2f} with the defaults"),
            (rf"\*\*{N} MP/s\*\* bf16 enqueue-only \({N} fp16, {N} synchronous\); decoder 512² level {N} ms per request at {N} GB/s",
             f"**{f['casc']:.2f} MP/s** bf16 enqueue-only ({f['casc16']:.2f} fp16, {f['casc_sync']:.2f} synchronous); decoder 512² level {f['d512']:.1f} ms per request at {f['d512_gbps']:.0f} GB/s"),
            (rf"default plan: {N} MP/s; 64 independent tiles: {N} MP/s; fp16 storage: {N} MP/s; exact-fp32 mode: {N} MP/s = {N}",
             f"default plan: {f['grid32']:.2f} MP/s; 64 independent tiles: {f['tiles']:.2f} MP/s; fp16 storage: {f['g16']:.2f} MP/s; exact-fp32 mode: {f['g32']:.2f} MP/s = {f['g32_frac']:.2f}"),
            (rf"TTFT {N} ms, TTST {N} ms", f"TTFT {f['ttft']:.1f} ms, TTST {f['ttst']:.1f} ms"),
            (rf"\*\*{N} passed, {N} skipped\*\* \(the two two-GPU tests\)", f"**{f['passed']} passed, {f['skipped']} skipped** (the two two-GPU tests)"),
            (rf"\(d40 4096²: {N} µs, {N} %; d64 {N} %; d128 {N} %; d160 {N} %\)", f"(d40 4096²: {f['a40_us']:.1f} µs, {f['a40']:.1f} %; d64 {f['a64']:.1f} %; d128 {f['a128']:.1f} %; d160 {f['a160']:.1f} %)"),
            (rf"\({N} ms per forward; `f2w` wide tile, `f6` the few-cout output conv at {N} µs\)", f"({f['dec_ms']:.2f} ms per forward; `f2w` wide tile, `f6` the few-cout output conv at {f['fc_us']:.0f} µs)"),
            (rf"`roofline\.single_lane\.avg_launch_us` \({N}\) is what the one-lane trace reproduces \({N} µs,", f"`roofline.single_lane.avg_launch_us` ({f['sl_us']:.1f}) is what the one-lane trace reproduces ({f['tr_us']:.1f} µs,"),
        ])
    for k in sorted(f):
        print(f"{k:14s} {f[k]}")


if __name__ == "__main__":
    main()
