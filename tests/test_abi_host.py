"""CPU: the C-ABI shared library loads and exports every symbol include/td_engine.h declares; host-side logic
(scheduler mirror, tile geometry, conditioning vector) against the golden vectors.  No compute calls (no GPU here)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "td_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(td_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    import ctypes
    from terrain_diffusion_amd._lib import LIB_PATH, EXPORTS
    lib = ctypes.CDLL(LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/td_engine.h but not exported"
    assert set(EXPORTS) == set(declared), set(EXPORTS) ^ set(declared)


def test_a_c_host_links_the_engine_library_and_gets_no_cpu_fallback(tmp_path):
    """tests/engine_host.c (C99) against libtd_engine.so: version / build id / _tile_seed answer on the host; without a GPU td_engine_create
    returns a negative code and a message instead of an engine."""
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    import terrain_diffusion_amd as td
    from terrain_diffusion_amd._lib import LIB_PATH
    libdir = os.path.dirname(LIB_PATH)
    exe = str(tmp_path / "engine_host")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "engine_host.c"), "-o", exe,
                           "-L", libdir, "-ltd_engine", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines())
    assert lines["build"] == ge.csrc_sha16() and int(lines["version"]) >= 1
    assert int(lines["seed"]) == td._tile_seed(5861, -3, 7)
    if not torch.cuda.is_available():
        code, msg = lines["engine_create"].split(":", 1)
        assert int(code) < 0 and len(msg.strip()) > 8, lines


def test_no_cpu_fallback_is_loud():
    """without a GPU the engine refuses to start instead of silently computing on the CPU."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import terrain_diffusion_amd as td
    with pytest.raises((td.TdError, RuntimeError)):
        td.EDMUnet2D(image_size=512, in_channels=5, model_channels=64, conditional_inputs=[["tensor", 58, 1.0]], fourier_scale="pos")
    with pytest.raises(RuntimeError):
        from terrain_diffusion_amd.engine import get_engine
        get_engine("cpu")


def test_tile_seed_host_function(golden):
    import terrain_diffusion_amd as td
    g = golden("rng")
    for (seed, _, _), (ty, tx), ref in zip(g["tile_seed_in"], g["tile_seed_in_signed"], g["tile_seed_out"]):
        assert td._tile_seed(int(seed), int(ty), int(tx)) == int(ref)


def test_scheduler_mirror_matches_reference_trace(golden):
    """the Python EDMDPMSolverMultistepScheduler mirror (host maths, torch CPU) reproduces the reference's step() trace."""
    import terrain_diffusion_amd as td
    from oracle import rng
    g = golden("schedule")
    for n in (4, 12, 20, 32):
        sch = td.EDMDPMSolverMultistepScheduler(sigma_min=0.002, sigma_max=80.0, sigma_data=0.5)
        sch.set_timesteps(n)
        assert np.array_equal(sch.sigmas.numpy(), g[f"sigmas_{n}"])
        x = torch.from_numpy(rng.standard_normal(900 + n, (2, 5, 8, 8))) * sch.sigmas[0]
        for i, (t, sigma) in enumerate(zip(sch.timesteps, sch.sigmas)):
            xin = sch.precondition_inputs(x, sigma)
            cn = sch.trigflow_precondition_noise(sigma.view(-1))
            F_ = torch.tanh(0.3 * xin) - 0.2 * torch.cos(cn)
            x = sch.step(F_, t, x).prev_sample
            err = float((x - torch.from_numpy(g[f"trace_{n}"][i])).pow(2).mean().sqrt() / torch.from_numpy(g[f"trace_{n}"][i]).pow(2).mean().sqrt())
            assert err < 2e-6, (n, i, err)


def test_host_geometry_and_conditioning(golden):
    import terrain_diffusion_amd as td
    from oracle import rng
    g = golden("geometry")
    flat, pos = g["tile_starts_flat"], 0
    for case, n in zip(g["tile_starts_cases"], g["tile_starts_len"]):
        assert td._tile_starts(*[int(v) for v in case]) == [int(v) for v in flat[pos:pos + n]]
        pos += n
    gs = golden("sampling")
    cond_img = torch.from_numpy(rng.standard_normal(31, (2, 7, 4, 4)))
    means = torch.tensor([0.1, -0.2, 0.3, 0.0, 1.0, -1.0, 0.0])
    stds = torch.tensor([1.0, 2.0, 0.5, 1.5, 1.0, 3.0, 1.0])
    got = td._process_cond_img(cond_img, torch.tensor([[0.1, 0.2, 0.3, 0.4, 0.5]]), means, stds, torch.full((2,), 0.25))
    assert np.allclose(got.numpy(), gs["cond58"], rtol=1e-6, atol=1e-6)


def test_committed_bench_line_honours_the_contract():
    """profiles/r04_bench_grid8.json is the line `python bench.py` (default: BASELINE configs[2]) printed on the MI355X: every field the driver / judge reads is there,
    the roofline fraction is consistent with its parts, the CPU baseline is labelled as the port it is, the HBM traffic comes from a counter file
    stamped with the build id of the library that ran (round 3), the strong-scaling anchor and single-tile leg are present, and (round 4) the
    small-batch kernel family is reported beside the roofline kernel."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "r04_bench_grid8.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    pj = json.load(open(os.path.join(root, "profiles", "r04_hbm_traffic_and_mfma_util.json")))
    assert d["roofline"]["library_build_id"] == pj["library_build_id"] == pj["csrc_sha16"] and d["roofline"]["traffic_source"].endswith("r04_hbm_traffic_and_mfma_util.json")
    assert d["roofline"]["traffic"] > d["roofline"]["flop_per_launch"] / 2500e12 * 0   # present and positive
    assert d["strong_scaling_anchor"]["value"] > 0 and "configs[3]" in d["strong_scaling_anchor"]["workload"]
    assert d["latency_single_tile_ms"] > 0 and d["roofline_single_tile"]["bound"] == "hbm"
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "MP/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and "configs[2]" in d["config"]["workload"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["flop_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12) / r["achieved"] < 0.01   # flop per launch / avg launch time
    assert r["traffic"] is None or r["traffic"] > 0
    sbk = r["small_batch_kernel"]   # round 4: the launches whose grid does not fill the chip (8x8 level at batch 64) run on conv_sb and are not in the roofline family
    assert sbk["kernel"] == "td::conv_sb_kernel" and sbk["launches_per_step"] + r["launches_per_step"] == r["all_conv_launches_per_step"]
    assert d["latency_single_tile_ms"] < 30.0   # round 4: the single-tile leg runs on the small-batch flavour (34.6 ms before)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "MP/s" and c["cores"] >= 1 and "sample" in c
    # value = whole-job MP per second: tiles x 0.262144 MP / step time
    assert abs(d["value"] - d["config"]["decoded_mp_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3


@pytest.mark.parametrize("rnd", ["r05", "r06"])
def test_stamped_bench_line_belongs_to_its_sources(rnd):
    """profiles/rNN_bench_grid8.json (the stamped line of a round): its counter file carries the build id of the library that ran; for the NEWEST round
    that id is the hash of the kernel sources in THIS tree (csrc/* + include/td_engine.h) -- a later edit of the kernels without a re-collection fails
    here (the round-5 line stays as a record: its ids only have to agree with each other, the round-6 kernels are not the ones it measured); the
    roofline fraction follows from flop_per_launch and the live launch time, the rocprofv3 trace of the same command agrees with it, and the two
    traffic ratios (with / without the optional second output counted as algorithmic) are what the counters and the labels give."""
    import csv
    import json
    import __graft_entry__ as ge
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "profiles", f"{rnd}_bench_grid8.json")):
        pytest.skip(f"no stamped {rnd} collection in profiles/ yet")
    d = json.loads(open(os.path.join(root, "profiles", f"{rnd}_bench_grid8.json")).read().strip().splitlines()[-1])
    pj = json.load(open(os.path.join(root, "profiles", f"{rnd}_hbm_traffic_and_mfma_util.json")))
    r = d["roofline"]
    assert r["library_build_id"] == pj["library_build_id"] == pj["csrc_sha16"]
    if rnd == "r06":
        assert pj["csrc_sha16"] == ge.csrc_sha16(), "the kernels were edited after the round-6 collection: re-run tools/r06_final.sh"
    assert "configs[2]" in d["config"]["workload"] and d["dtype"] == "bf16" and d["vs_baseline"] is None and d["n_gpus"] == 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["peak"] == 2500.0 and r["bound"] == "mfma"
    assert abs(r["achieved"] - r["flop_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12) / r["achieved"] < 0.01
    # the kernel trace: call-weighted average launch time of the conv_glds instantiations (the wide tile's kernel name contains the family's) within 3 % of
    # the live one.  Round 6: the bench line runs two sampler lanes whose kernels overlap; the trace is taken on ONE lane (`--engine-opts dual_stream=0`)
    # and reproduces the line's `roofline.single_lane` leg (same launches at the full batch, timed one by one with HIP events)
    live = r["single_lane"] if r.get("lanes", 1) == 2 else r
    if r.get("lanes", 1) == 2:
        assert abs(live["frac"] - live["achieved"] / r["peak"]) < 1e-3
        assert abs(live["achieved"] - r["flop_per_launch"] / (live["avg_launch_us"] * 1e-6) / 1e12) / live["achieved"] < 0.01
    rows = [x for x in csv.DictReader(open(os.path.join(root, "profiles", f"{rnd}_bench_grid8_kernel_trace_summary.csv"))) if r["kernel"].split("::")[-1].split(" ")[0] in x["kernel"]]
    calls, total = sum(int(x["calls"]) for x in rows), sum(float(x["total_us"]) for x in rows)
    assert calls % r["launches_per_step"] == 0 and abs(total / calls - live["avg_launch_us"]) / live["avg_launch_us"] < 0.03
    # traffic: measured bytes per launch over the algorithmic bytes, with and without the pre-activated second output
    assert r["traffic_algorithmic_strict"] < r["traffic_algorithmic"] < r["traffic"]
    assert abs(r["traffic_over_algorithmic"] - r["traffic"] / r["traffic_algorithmic"]) < 2e-3
    assert abs(r["traffic_over_algorithmic_strict"] - r["traffic"] / r["traffic_algorithmic_strict"]) < 2e-3
    assert abs(d["value"] - d["config"]["decoded_mp_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "MP/s" and c["cores"] >= 1 and "sample" in c
    assert d["roofline_single_tile"]["bound"] == "hbm" and 15.0 < d["latency_single_tile_ms"] < 25.0


def test_unsupported_constructor_arguments_are_refused_before_the_engine_starts():
    """reference options the accelerated path does not carry are refused loudly (never silently remapped): block_kwargs / encode_only,
    fourier_scale != 'pos', 'embedding' conditional inputs, and noise_emb_dims=0 (edm_unet.py:49: 0 disables the noise input)."""
    import terrain_diffusion_amd as td
    base = dict(image_size=64, in_channels=5, model_channels=64, conditional_inputs=[["tensor", 58, 1.0]], fourier_scale="pos")
    for bad in (dict(noise_emb_dims=0), dict(fourier_scale=1), dict(encode_only=True), dict(conditional_inputs=[["embedding", 10, 1.0]])):
        with pytest.raises(NotImplementedError):
            td.EDMUnet2D(**{**base, **bad})
