#!/usr/bin/env python3
"""bench.py — composited frames/s of the scene rasteriser hot path on MI355X.

Workload (BASELINE.json metric "8x1080p -> 1 4K scene", configs[2]): 8 planar YUV420 1080p inputs,
Tiles{8 x View{Rescaler{InputStream} border_radius 24, Text label}} -> one 3840x2160 YUV420 frame.
One step = one composited output frame, inputs already resident in HBM, output left in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1: launched by torchrun (one rank per GPU); inputs are sharded across ranks, each rank
  resamples its inputs to tiles, tiles are gathered on rank 0 over RCCL, rank 0 composes
  (strong scaling of one scene: total work fixed).

Prints ONE JSON line on rank 0 (see README / DESIGN.md §measurement for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

IN_W, IN_H, OUT_W, OUT_H, N_IN = 1920, 1080, 3840, 2160, 8
RING = int(os.environ.get("SMR_BENCH_RING", "12"))  # distinct input frame sets cycled through, 12 x 24.9 MB > the 256 MB Infinity Cache (the env: diagnostics only)


def yuv420_bytes(w, h):
    return w * h * 3 // 2


ALGO_BYTES_PER_FRAME = N_IN * yuv420_bytes(IN_W, IN_H) + yuv420_bytes(OUT_W, OUT_H)  # 37 324 800


LABEL_W, LABEL_H = 176, 32
PLAIN_TILES = False  # --config 1: Tiles of bare input streams (rescale + blend only)
ANIMATED = False     # --config 4: animated Tiles grid + one gaussian-blur layer (smelter_amd/synth.py:animated_grid_scene)
LAYER_W, LAYER_H, LAYER_SIGMA = 960, 540, 3.0
UPDATE_EVERY = 45    # --config 4: a scene update (tiles swap places, 500 ms transition) every 45 frames = 0.75 s at 60 fps


def scene_json(rotate=0):
    """configs[2] as the scene JSON the reference's API takes (smelter-api/src/video/component.rs)."""
    if ANIMATED:
        from smelter_amd import synth
        return synth.animated_grid_scene(N_IN, rotate, LAYER_W, LAYER_H, 500, LAYER_SIGMA)
    if PLAIN_TILES:
        return {"type": "tiles", "background_color": "#000000FF", "children": [{"type": "input_stream", "input_id": f"input_{i}"} for i in range(N_IN)]}
    kids = []
    for i in range(N_IN):
        label = {"type": "view", "background_color": "#00000080", "border_radius": 8.0, "width": float(LABEL_W), "height": float(LABEL_H),
                 "left": 24.0, "bottom": 24.0, "padding_vertical": 4.0, "padding_horizontal": 8.0,
                 "children": [{"type": "text", "text": "CAM 3 LIVE", "font_size": 24.0, "width": float(LABEL_W), "height": float(LABEL_H)}]}
        kids.append({"type": "view", "background_color": "#101018FF",
                     "children": [{"type": "rescaler", "border_radius": 24.0, "child": {"type": "input_stream", "input_id": f"input_{i}"}}, label]})
    return {"type": "tiles", "background_color": "#202030FF", "children": kids}


def build_scene():
    """Scene JSON -> flattened layout list + per-source resolutions through the host scene engine
    (smelter_amd/csrc/host/scene*.cpp; outside the timed region: the scene does not change between frames)."""
    from smelter_amd import _ffi
    from smelter_amd.scene import Scene
    sc = Scene()
    nodes = sc.update(scene_json(), OUT_W, OUT_H)
    res = [(IN_W, IN_H) if nodes[k].kind == _ffi.NODE_INPUT_STREAM else (nodes[k].width, nodes[k].height) for k in nodes[0].children]
    global INNER_LAYOUTS
    if ANIMATED:  # the layout node under the blur shader: View{Rescaler{input_0}} at the layer's size
        shader = list(nodes[0].children)[-1]
        INNER_LAYOUTS = sc.layouts(list(nodes[shader].children)[0], 0, [(IN_W, IN_H)])
    return sc.layouts(0, 0, res), res


INNER_LAYOUTS = None


def make_inputs(ctx, hip, frame_sets, input_ids):
    """frame_sets x len(input_ids) device frames of synthetic 1080p YUV420 (TestInput pattern + seeded noise + shift)."""
    from smelter_amd import synth
    ring = []
    for s in range(frame_sets):
        row = {}
        for i in input_ids:
            y, u, v = synth.test_input(i, IN_W, IN_H, noise_seed=1234 + i + 100 * s, shift=s * 7)
            row[i] = ctx.frame(hip.FRAME_PLANAR_YUV420, IN_W, IN_H, [y, u, v])
        ring.append(row)
    return ring


_BOOK = [None, False]


def font_book():
    """The library's own text pipeline (smr_fontbook_*: C++ TrueType reader, layout, rasteriser) over the machine's TrueType fonts; None where
    there are none (the labels then come from synth.label_glyphs' procedural 5x7 font — same node size, same per-frame work)."""
    if not _BOOK[1]:
        _BOOK[1] = True
        try:
            from smelter_amd import fontbook as T
            _BOOK[0] = T.NativeFontBook.system()
        except Exception:
            _BOOK[0] = None
    return _BOOK[0]


def label_run():
    """(atlas, glyphs) of a tile's label: the Text node of scene_json() — "CAM 3 LIVE", 24 px, fixed 176 x 32 — as the font book rasterises it."""
    book = font_book()
    if book is None:
        from smelter_amd import synth
        return synth.label_glyphs("CAM 3 LIVE", 3)
    glyphs, atlas = book.rasterise("CAM 3 LIVE", LABEL_W, LABEL_H, 24.0)
    return atlas, glyphs


def make_label(ctx):
    atlas, glyphs = label_run()
    t = ctx.surface(LABEL_W, LABEL_H)
    ctx.blit_glyphs(t, (0.0, 0.0, 0.0, 0.0), glyphs, atlas)  # transparent background (text_renderer.rs default)
    return t


def cpu_baseline(layouts, res):
    """The CPU restatement of the reference renderer (oracle, kind 'port') on the same workload: about ten seconds of whole
    composited frames (all passes) with OpenMP on all host cores, plus one frame on a single thread."""
    from tests import refpipe
    from oracle import oracle as orc
    from smelter_amd import synth
    orc.build()
    planes = [synth.test_input(i, IN_W, IN_H, noise_seed=1234 + i) for i in range(N_IN)]
    atlas, glyphs = label_run()
    label = orc.blit_glyphs(LABEL_W, LABEL_H, orc.color_to_shader((0, 0, 0, 0), True), glyphs, atlas)
    cores = orc.num_threads(omp=True)

    def one_frame(omp=True):
        if not ANIMATED:
            # the whole frame in one C call (oracle/smr_oracle.c:orc_render_frame_yuv420): no Python between the passes
            sources, k = [], 0
            for r in res:
                if r == (IN_W, IN_H):
                    sources.append(k); k += 1
                else:
                    sources.append(label)
            orc.render_frame_yuv420(planes, layouts, sources, OUT_W, OUT_H, omp=omp)
            return
        nodes, k = [], 0
        for r in res:
            if r == (IN_W, IN_H):
                y, u, v = planes[k]
                k += 1
                nodes.append(orc.planar_yuv_to_rgba(y, u, v, IN_W, IN_H, omp=omp))
            else:  # the blur layer: its own layout node, then the shader
                inner = refpipe.layout_node_render(INNER_LAYOUTS, [nodes[0]], LAYER_W, LAYER_H, omp=omp)
                nodes.append(orc.gaussian_blur(inner, LAYER_SIGMA))
        refpipe.render_yuv420(layouts, nodes, OUT_W, OUT_H, omp=omp)

    # a bounded sample of about ten seconds of CPU work: one frame to size it, then as many frames as fit
    t0 = time.perf_counter()
    one_frame()
    first = time.perf_counter() - t0
    reps = int(min(max(12.0 / first, 1), 200))
    per_frame = []
    for _ in range(reps):
        t0 = time.perf_counter()
        one_frame()
        per_frame.append(time.perf_counter() - t0)
    dt = sum(per_frame) / reps
    t0 = time.perf_counter()
    one_frame(omp=False)  # all host cores and one core
    dt1 = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "spread": {"min_frames_per_s": round(1.0 / max(per_frame), 4), "max_frames_per_s": round(1.0 / min(per_frame), 4), "frames": reps,
                       "what": "slowest and fastest single frame of the sample (the box's host cores are shared: the figure moves between runs)"},
            "context": "a plain restatement of the reference's passes (test oracle), not a tuned CPU renderer: the GPU / CPU ratio says nothing about "
                       "kernel quality (the roofline fraction does).  The reference's own software path (wgpu on lavapipe) is published at 60 composited "
                       "1080p frames/s on 16 vCPU (c5.4xlarge; benchmarks/2025_04_28_9891af76/full_c5.4xlarge.json:607-610, BASELINE.md row 1) — "
                       "another workload and machine, quoted for scale only",
            "one_core": {"value": round(1.0 / dt1, 4), "unit": "frames/s", "sample": f"1 frame, single thread, {dt1:.1f} s"},
            "sample": f"{reps} composited frames of the same workload ({N_IN}x{IN_W}x{IN_H} YUV420 -> {OUT_W}x{OUT_H} YUV420, all passes; "
                      f"first frame {first:.2f} s discarded as warm-up), oracle/smr_oracle.c all-C frame loop (-O3 -mavx2 -mfma, "
                      f"OpenMP over rows) on {cores} threads, {dt:.3f} s per frame"}


class Watchdog:
    """The N > 1 path's first contact with hardware must not cost the driver its whole slot: every step and every barrier beats; a rank that
    makes no progress for `seconds` (a stuck smr_gather_tiles / RCCL send-recv pair, a peer that died) prints an error — rank 0 as the ONE JSON
    line, with "error" and a null value — and ends the process with rc 3 (torchrun then ends the other ranks)."""

    def __init__(self, seconds, rank, world, config):
        import threading
        self.seconds, self.rank, self.world, self.config = seconds, rank, world, config
        self.t, self.where, self.on = time.monotonic(), "start", seconds > 0
        if self.on:
            threading.Thread(target=self._run, daemon=True).start()

    def beat(self, where):
        self.t, self.where = time.monotonic(), where

    def stop(self):
        self.on = False

    def _run(self):
        while self.on:
            time.sleep(0.5)
            idle = time.monotonic() - self.t
            if self.on and idle > self.seconds:
                msg = f"watchdog: rank {self.rank} of {self.world} made no progress for {idle:.0f} s in `{self.where}`"
                print("[bench] " + msg, file=sys.stderr, flush=True)
                if self.rank == 0:
                    print(json.dumps({"metric": "composited frames/sec", "value": None, "unit": "frames/s", "n_gpus": self.world, "error": msg,
                                      "config": {"workload": f"configs[{self.config}] sharded"}, "higher_is_better": True}), flush=True)
                os._exit(3)


def quoted_profile(name):
    """A committed counter profile (profiles/<name>) — quoted only if it was collected on the library being timed: the file records the
    sha256 of that library's device code (tools/traffic_json.py, tools/issue_json.py <- smelter_amd/build.py:kernels_sha256)."""
    from smelter_amd import _ffi, build as hip_build
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, {"stale": True, "why": f"profiles/{name} does not exist"}
    data = json.load(open(path))
    want = (data.get("_identity") or {}).get("lib_kernels_sha256")
    have = hip_build.kernels_sha256(_ffi.LIB_PATH)
    if want != have:
        return None, {"stale": True, "why": f"profiles/{name} was collected on device code {str(want)[:12]}, this library is {have[:12]}"}
    return data, {"stale": False, "lib_kernels_sha256": have}


def roofline_block(args, stages, kernel_bytes, knames, ms_per_step, algo_bytes=None):
    """The HBM roofline entry of the line.  The path spans three launches per frame, so "the kernel" of the contract is ambiguous: the
    figure of ONE launch rises whenever work is split across more launches (round 4: 0.087 -> 0.139 by splitting wave A).  The headline
    therefore prices the frame's algorithmic bytes (inputs read once in their native format + output written once)
    against ALL the launches of a frame — the sum of their mean durations (HIP events on the ctx stream, one frame in flight) — which no
    split can move; the single dominant launch (the contract's literal reading) and the pipelined frame period are reported beside it."""
    if not stages:
        return None
    ALGO_BYTES_PER_FRAME = algo_bytes if algo_bytes is not None else globals()["ALGO_BYTES_PER_FRAME"]
    dom = max(stages, key=lambda k: stages[k]["avg_us"])
    sum_us = sum(v["avg_us"] for v in stages.values())
    dom_us = stages[dom]["avg_us"]
    ach = ALGO_BYTES_PER_FRAME / (sum_us * 1e-6) / 1e9
    traffic, ident = None, None
    tname = {2: "r06_traffic.json", 3: "r06_traffic_configs3.json"}.get(args.config)  # the committed counter passes: default workload, target
    per_kernel_traffic = None
    if tname and args.ingest == "auto":
        data, ident = quoted_profile(tname)
        if data:
            per_kernel_traffic = {knames[k]: data[knames[k]]["hbm_bytes_per_launch"] for k in stages if knames.get(k) in data}
            traffic = sum(per_kernel_traffic.values()) if len(per_kernel_traffic) == len(stages) else None
    wave_a = [k for k in ("ingest", "fused_ingest_resample") if k in stages]
    block = {"bound": "hbm", "kernel": " + ".join(knames.get(k, k) for k in stages), "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBPS, 5), "bytes_per_launch": ALGO_BYTES_PER_FRAME, "avg_launch_us": round(sum_us, 3),
             "definition": "the frame's algorithmic bytes / the sum of the mean launch durations of the frame's kernels (one frame in flight, HIP events)",
             "traffic": traffic, "traffic_per_kernel": per_kernel_traffic, "traffic_stale": bool(ident and ident.get("stale")),
             "traffic_source": (f"profiles/{tname} (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, separate passes; device code {ident['lib_kernels_sha256'][:12]})"
                                if traffic is not None else (ident or {}).get("why")),
             "traffic_over_algorithmic": round(traffic / ALGO_BYTES_PER_FRAME, 3) if traffic else None,
             "dominant_kernel": {"kernel": knames.get(dom, dom), "avg_launch_us": dom_us,
                                 "achieved": round(ALGO_BYTES_PER_FRAME / (dom_us * 1e-6) / 1e9, 2),
                                 "frac": round(ALGO_BYTES_PER_FRAME / (dom_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5),
                                 "own_bytes": kernel_bytes.get(dom), "own_GBps": round(kernel_bytes.get(dom, 0) / (dom_us * 1e-6) / 1e9, 2),
                                 "what": "the contract's literal reading (frame bytes / the longest single launch); own_* = that launch's own inputs + outputs"},
             "pipelined_frame": {"us": round(ms_per_step * 1e3, 3), "frac": round(ALGO_BYTES_PER_FRAME / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                                 "what": "frame bytes / ms_per_step of the timed loop (frames in flight overlap)"},
             "wave_a_us": round(sum(stages[k]["avg_us"] for k in wave_a), 3),
             "limiter": ("latency under memory load, between the two roofs: alone, every kernel runs at 40 - 80 % of its vector-issue rate and a third of the copy bandwidth; with two "
                         "frames in flight the frame period follows the WAITS, not the instruction or byte counts — a converter without its per-pixel arithmetic (- 55 % of its "
                         "instructions) gains 2 %, one without its stores 3 %, while deferring the resampler's tile stores behind the next chunk's wait and turning the converter's "
                         "and compositor's FLAT accesses into GLOBAL ones (counted waits instead of vmcnt(0)) gained 6.5 % with no instruction or byte removed "
                         "(profiles/r06_sensitivity.txt).  `issue_frac` prices the frame's instructions at the guide's peak issue rate, `frac` its algorithmic bytes at 8 TB/s: "
                         "both far from 1 — DESIGN.md section 3")
             if args.ingest == "auto" else "see DESIGN.md section 3"}
    if args.config == 2 and args.ingest == "auto":
        iss, ident2 = quoted_profile("r06_issue.json")
        if iss:
            fl = iss["floors_us_per_frame"]
            guide = min(v["floor_us_pipes_overlapped"] for v in fl.values())
            block["issue_frac"] = round(guide / (ms_per_step * 1e3), 4)  # the instruction-issue roof beside the HBM one: floor at the guide's peak rate / pipelined frame time
            block["issue"] = {"valu_wave_instructions_per_frame": iss["per_frame"]["valu_wave_instructions"],
                              "mfma_wave_instructions_per_frame": iss["per_frame"]["mfma_wave_instructions"],
                              "floors_us_per_frame": {k: v["floor_us_pipes_overlapped"] for k, v in fl.items()},
                              "frame_us": round(ms_per_step * 1e3, 2),
                              "frame_over_floor": {k: round(ms_per_step * 1e3 / v["floor_us_pipes_overlapped"], 2) for k, v in fl.items()},
                              "issue_frac": {k: round(v["floor_us_pipes_overlapped"] / (ms_per_step * 1e3), 4) for k, v in fl.items()},
                              "reading": iss.get("reading"),
                              "source": "profiles/r06_issue.json (rocprofv3 --pmc SQ_INSTS_VALU / SQ_INSTS_MFMA of this command on this device code; vector and "
                                        "matrix pipes overlap: floor = the larger)"}
        else:
            block["issue"] = {"stale": True, "why": ident2.get("why")}
    return block


def capacity_mode(args):
    """The reference's own capacity benchmark, "rendering only" set (integration-tests/src/bin/benchmark/suite.rs:332-346, scenes.rs:293-327,
    benchmark_pass.rs:331-409): ONE raw YUV420 input, uploaded from host memory every frame, feeds N outputs — each its own Tiles scene
    (margin 2, grey background) at the input's resolution — and every output frame is read back to host memory; the result is the largest N
    the renderer sustains at the frame rate (the reference checks pts progress after 6 s; here: the measured time per frame of N outputs,
    upload and read-back included, must stay within 1 / fps over a 6 s run).  The only published numbers for this path are of this shape
    (T4 N = 64, c5.4xlarge N = 2 at 1080p30) — different hardware, reported beside, never as `vs_baseline`."""
    from smelter_amd import hip
    from smelter_amd.renderer import Renderer
    w, h, fps = args.capacity_width, args.capacity_height, args.capacity_fps
    budget = 1.0 / fps
    scene = {"type": "tiles", "margin": 2.0, "background_color": "#808080FF", "children": [{"type": "input_stream", "input_id": "input_0"}]}
    rng = np.random.default_rng(11)
    probes = []

    def seconds_per_frame(n, seconds):
        ctx = hip.Context(0)
        r = Renderer(ctx, stream_fallback_timeout_s=3600.0, max_outputs=n)
        r.register_input("input_0")
        for i in range(n):
            r.update_scene(f"out_{i}", w, h, scene)
        ring = 4  # host-side frames of the "decoder": pinned, refreshed round-robin
        dev_in = ctx.frame(hip.FRAME_PLANAR_YUV420, w, h)
        host_in = []
        for k in range(ring):
            planes = dev_in.pinned_planes()
            for p_ in planes:
                p_[...] = rng.integers(16, 236, p_.shape, dtype=np.uint8)
            host_in.append(planes)
        host_out = None
        fs = r.make_frame_set({"input_0": dev_in})

        def one(k):
            nonlocal host_out
            dev_in.upload_async(host_in[k % ring])
            cnt = r.render_packed(k * (1_000_000_000 // fps), fs)
            if host_out is None:
                host_out = [r.output(i).pinned_planes() for i in range(cnt)]
            for i in range(cnt):
                r.output(i).download_async(host_out[i])
            ctx.sync()  # the frame's outputs are in host memory: one frame in flight, like the reference's render thread
            return cnt
        for k in range(3):
            got = one(k)
        assert got == n, (got, n)
        t0 = time.perf_counter()
        k = 0
        while True:
            one(3 + k)
            k += 1
            if time.perf_counter() - t0 >= seconds:
                break
        dt = (time.perf_counter() - t0) / k
        r.close()
        ctx.close()
        return dt, k

    def holds(n, seconds):
        dt, frames = seconds_per_frame(n, seconds)
        probes.append({"outputs": n, "ms_per_frame": round(dt * 1e3, 3), "frames": frames, "seconds": seconds, "holds": dt <= budget})
        print(f"[bench] capacity: N = {n}: {dt * 1e3:.2f} ms per frame ({'holds' if dt <= budget else 'fails'} {fps} fps)", file=sys.stderr, flush=True)
        return dt <= budget
    lo, hi = 0, 8
    while hi <= args.capacity_max and holds(hi, 1.0):
        lo, hi = hi, hi * 2
    hi = min(hi, args.capacity_max + 1)
    while hi - lo > max(1, lo // 32):  # (to ~3 %: every probe builds N outputs)
        mid = (lo + hi) // 2
        if holds(mid, 1.0):
            lo = mid
        else:
            hi = mid
    while lo > 0 and not holds(lo, 6.0):  # the reference's first check: 6 s
        lo = max(lo - max(1, lo // 32), 0)
    final = [p_ for p_ in probes if p_["outputs"] == lo and p_["seconds"] == 6.0]
    print(json.dumps({"metric": f"capacity, rendering only: largest N with 1 raw {w}x{h} YUV420 input (uploaded per frame) -> N Tiles outputs (each read back) at {fps} fps",
                      "value": lo, "unit": "outputs", "n_gpus": 1, "higher_is_better": True, "frames_per_s_equivalent": lo * fps,
                      "ms_per_frame_at_value": final[-1]["ms_per_frame"] if final else None, "budget_ms": round(budget * 1e3, 3),
                      "data": "synthetic (random limited-range YUV420 frames in pinned host memory, a ring of 4)",
                      "config": {"workload": "integration-tests benchmark 'rendering only' / tiles_1_to_n (suite.rs:332-346, scenes.rs:293-327): one frame in flight, "
                                             "upload + N x (resample + compose + read-back) per frame", "resolution": [w, h], "fps": fps},
                      "reference_published_other_hardware": {"g4dn.xlarge (NVIDIA T4), 1080p30": 64, "g4dn.2xlarge (T4), 1080p30": 67, "c5.4xlarge (16 vCPU, software Vulkan), 1080p30": 2,
                                                             "source": "BASELINE.md rows 'Tiles scene, 1080p30'"},
                      "vs_baseline": None, "probes": probes}))


def same_device_mode(args):
    """The N > 1 code path as far as ONE GPU can run it: `--same-device N` drives N ShardedCompositors — rank r = its own context (HIP stream) on
    device 0 — through the pipelined protocol of `--gpus N` (ingest k on every rank, gather k posted, root composes k - 1) over a local
    communicator (smr_comm_create_local: device-to-device copies + events instead of RCCL over xGMI).  It exercises the sharded driver, the
    double-buffered tile sets and the stream ordering on hardware; it says NOTHING about multi-GPU scaling — all ranks share one GPU's CUs and
    HBM, so the value is expected near the one-GPU figure.  Informational; the scaling curve (SCALE_rNN.json) is the driver's."""
    global IN_W, IN_H, OUT_W, OUT_H, N_IN, ALGO_BYTES_PER_FRAME
    IN_W, IN_H, OUT_W, OUT_H, N_IN = 3840, 2160, 3840, 2160, 8  # configs[3]
    ALGO_BYTES_PER_FRAME = N_IN * yuv420_bytes(IN_W, IN_H) + yuv420_bytes(OUT_W, OUT_H)
    import torch
    from smelter_amd import dist as smr_dist
    from smelter_amd import hip
    world = args.same_device
    ctxs = [hip.Context(0) for _ in range(world)]
    layouts, res = build_scene()
    label = make_label(ctxs[0])
    plan = smr_dist.ShardPlan(n_inputs=N_IN, world=world)
    RING_SD = 3
    rings = [make_inputs(ctxs[r], hip, RING_SD, plan.inputs_of(r)) for r in range(world)]
    slots = [i for i, r in enumerate(res) if r == (IN_W, IN_H)]
    comm = hip.Comm.local(ctxs)
    ranks = smr_dist.LocalRanks(comm)
    sc = [smr_dist.ShardedCompositor(ctxs[r], hip, plan, r, layouts, res, slots, label if r == plan.root else None, torch, None, comm=ranks.view(r)) for r in range(world)]
    outs = [ctxs[plan.root].frame(hip.FRAME_PLANAR_YUV420, OUT_W, OUT_H) for _ in range(2)]

    def step(k):
        for r in ranks.order(plan.root):
            sc[r].step_pipelined(rings[r][k % RING_SD], outs[k % 2] if r == plan.root else None)

    def barrier():
        for r in ranks.order(plan.root):
            sc[r].flush()
        for c in ctxs:
            c.sync()
    for k in range(4 + args.warmup):
        step(k)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    barrier()
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "composited frames/sec, 8x4K -> 4K sharded over N contexts of ONE device (code-path evidence, not a scaling point)", "value": round(args.steps / dt, 2),
                      "unit": "frames/s", "n_gpus": 1, "ranks_on_one_device": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
                      "higher_is_better": True, "data": "synthetic", "dtype": "u8 in / u8 out (f32 conversion, f16-pair matrix-core resampler)",
                      "config": {"workload": "configs[3]: 8x4K YUV420 inputs, input i on rank i mod N, tiles gathered by device-to-device copies (smr_comm_create_local), root composes frame k - 1 while frame k travels",
                                 "transport": "hipMemcpy2DAsync + events on one device (RCCL is the transport of --gpus N)"},
                      "vs_baseline": None}))
    comm.close()
    for c in ctxs:
        c.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--same-device", type=int, default=0,
                    help="N > 1: the sharded (multi-GPU) driver with N ranks as N contexts of device 0 over a local communicator — the code path of --gpus N on one GPU "
                         "(informational: no scaling can be read from it)")
    ap.add_argument("--mode", choices=["throughput", "capacity"], default="throughput",
                    help="throughput (default): the judged line — composited frames/s of configs[2] with inputs resident in HBM.  capacity: the reference's own "
                         "'rendering only' capacity benchmark shape (1 raw input uploaded per frame -> N outputs read back, largest N at the frame rate)")
    ap.add_argument("--capacity-width", type=int, default=1920)
    ap.add_argument("--capacity-height", type=int, default=1080)
    ap.add_argument("--capacity-fps", type=int, default=30)
    ap.add_argument("--capacity-max", type=int, default=2048)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-frames", type=int, default=2000)
    ap.add_argument("--ingest", choices=["auto", "valu", "fused", "mfma", "mfma_node"], default="auto",
                    help="wave A: exact converter into node textures + matrix-core resampler (auto, default; mfma / mfma_node are old names of it), "
                         "exact f32 kernel (valu), matrix-core kernel with the fused, one-code-per-stage conversion (fused: opt-in, A/B)")
    ap.add_argument("--convert", choices=["auto", "general", "block4x2"], default="auto", help="input converter kernels (SMR_OPT_CONVERT_IMPL; A/B)")
    ap.add_argument("--direct-output", action="store_true",
                    help="SMR_OPT_DIRECT_OUTPUT: the resampling kernel writes Y'CbCr for the compositor's copy tiles (A/B; default off)")
    ap.add_argument("--plane-source", action="store_true",
                    help="SMR_OPT_PLANE_SOURCE: the resampling kernel reads the frames' planes and converts exactly in the wave — no converter launch, no node "
                         "texture in memory (A/B; default off: slower, DESIGN.md section 3c)")
    ap.add_argument("--prime-seconds", type=float, default=0.25,
                    help="untimed frames before the W warm-up steps, for this long: the device's clocks, caches and the host's code paths in the state of a running "
                         "compositor (0: none)")
    ap.add_argument("--no-long", action="store_true", help="skip the `value_long` loop (profiling runs: keeps traces small)")
    ap.add_argument("--long-seconds", type=float, default=12.0,
                    help="length of the `value_long` loop — the same timed loop run right after `value`, BEFORE any CPU work, long enough for an outside "
                         "observer sampling GPU activity every few seconds to see it")
    ap.add_argument("--no-target", action="store_true", help="skip the child-process blocks: the north-star target (8x4K -> 4K on one GPU) and the sharded code path with one rank")
    ap.add_argument("--watchdog-seconds", type=float, default=60.0, help="N > 1 / --force-sharded: a step that makes no progress for this long ends the run with an error line and rc 3")
    ap.add_argument("--inflight", type=int, default=None, help="frames in flight on one GPU (renderer contexts / HIP streams).  Default 2: best with three kernels "
                                                              "per frame (configs[2] 19.3k / 17.6k frames/s at 2 / 3); --config 4 defaults to 3 (a frame in motion is "
                                                              "eleven launches, most of them small: 5.95k / 6.19k / 5.77k at 2 / 3 / 4 — profiles/r06_sensitivity.txt)")
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU code path (ingest per shard, gather, compose) with the ranks given")
    ap.add_argument("--transfers", action="store_true", help="also time host buffers in -> host buffers out (PCIe inclusive, informational)")
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3, 4],
                    help="BASELINE.json configs[] index: 2 = the metric's 8x1080p -> 4K (default on one GPU, the judged line); "
                         "3 = 8x4K -> 4K (default for --gpus N > 1: BASELINE's multi-GPU config, one input per GPU at N = 8); "
                         "1 = 4x1080p -> 1080p tiles, 3 = 8x4K -> 4K on one GPU, 4 = 16x1080p animated grid + blur layer "
                         "-> 4K on one GPU (informational)")
    args = ap.parse_args()
    if args.mode == "capacity":
        return capacity_mode(args)
    if args.same_device > 1:
        return same_device_mode(args)
    if args.config is None:
        # N > 1 shards BASELINE's multi-GPU workload (configs[3]: 8x4K, one input per GPU at N = 8): configs[2]'s 1080p inputs leave a GPU
        # 7 us of work per tile it then sends over one xGMI link for 24 us — link-bound beyond one GPU
        args.config = 3 if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.gpus > 1 or args.force_sharded) else 2
    global IN_W, IN_H, OUT_W, OUT_H, N_IN, ALGO_BYTES_PER_FRAME, PLAIN_TILES, ANIMATED
    if args.config == 1:
        IN_W, IN_H, OUT_W, OUT_H, N_IN, PLAIN_TILES = 1920, 1080, 1920, 1080, 4, True
    elif args.config == 3:
        IN_W, IN_H, OUT_W, OUT_H, N_IN = 3840, 2160, 3840, 2160, 8
    elif args.config == 4:
        N_IN, ANIMATED = 16, True
    ALGO_BYTES_PER_FRAME = N_IN * yuv420_bytes(IN_W, IN_H) + yuv420_bytes(OUT_W, OUT_H)
    if ANIMATED:
        ALGO_BYTES_PER_FRAME += 2 * LAYER_W * LAYER_H * 4  # the blur layer written and read once

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)

    from smelter_amd import build as hip_build
    if not os.path.exists(hip_build.LIB):
        raise SystemExit("libsmr_hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    from smelter_amd import dist as smr_dist
    from smelter_amd import hip

    # Multi-GPU: the renderer enqueues on the same (non-default) torch stream the RCCL send/recv calls are issued under, so a
    # tile is sent only after the ingest kernel that writes it and composed only after it has arrived.  (The default stream's
    # handle is 0, which smr_ctx_create reads as "create your own stream" — hence an explicit side stream.)
    single = world == 1 and not args.force_sharded  # --force-sharded: the N > 1 code path with world_size 1 (no exchange), for tests
    side = torch.cuda.Stream() if not single else None
    if side is not None:
        torch.cuda.set_stream(side)
    ctx = hip.Context(local_rank, stream=side.cuda_stream if side is not None else None)
    ingest_impl = {"auto": hip.INGEST_AUTO, "valu": hip.INGEST_VALU_F32, "mfma": hip.INGEST_MFMA_F16, "fused": hip.INGEST_MFMA_F16_FUSED,
                   "mfma_node": hip.INGEST_MFMA_F16_NODE}[args.ingest]
    convert_impl = {"auto": hip.CONVERT_AUTO, "general": hip.CONVERT_GENERAL, "block4x2": hip.CONVERT_BLOCK_4X2}[args.convert]
    ctx.set_ingest_impl(ingest_impl)
    ctx.set_convert_impl(convert_impl)
    ctx.set_direct_output(args.direct_output)
    ctx.set_plane_source(args.plane_source)
    layouts, res = build_scene()
    packed = hip.pack_layouts(layouts)
    label = make_label(ctx)

    plan = smr_dist.ShardPlan(n_inputs=N_IN, world=world)
    my_inputs = plan.inputs_of(rank)
    ring = make_inputs(ctx, hip, RING, my_inputs)
    if args.inflight is None:
        args.inflight = 3 if args.config == 4 else 2
    n_lanes = max(1, args.inflight) if single else 1
    # (single GPU: every renderer owns two alternating output frames; sharded path: the root's two)
    outs = [ctx.frame(hip.FRAME_PLANAR_YUV420, OUT_W, OUT_H) for _ in range(2)] if rank == 0 and not single else []

    def out_for(step):
        return outs[step % 2]
    input_source_slot = [i for i, r in enumerate(res) if r == (IN_W, IN_H)]  # source index of input k

    lanes = [ctx]
    if single:
        # The whole per-frame path of the reference's Renderer::render (state.rs:220-252) is inside a step: frame set ->
        # populate_inputs -> layout maths at this pts (scene engine) -> parameter pack -> ingest + compose kernels -> output frame.
        # Frames in flight: ONE renderer with `--inflight` lanes (smr_renderer_add_lane: extra contexts = HIP streams + scratch +
        # output frames).  Consecutive frames rotate through the lanes, so the latency-bound tail of one frame's compose kernel
        # overlaps the next frame's ingest kernel; the scene is one state (one update_scene call, one pts sequence).  The
        # strictly serial rate, the per-kernel timing and the latency come from a second, single-lane renderer.
        from smelter_amd import _ffi, synth
        from smelter_amd.renderer import Renderer
        lanes += [hip.Context(local_rank) for _ in range(n_lanes - 1)]
        for c in lanes[1:]:
            c.set_ingest_impl(ingest_impl)
            c.set_convert_impl(convert_impl)
            c.set_direct_output(args.direct_output)
            c.set_plane_source(args.plane_source)
        atlas, glyphs = label_run()

        def make_renderer(c, extra=()):
            r = Renderer(c, stream_fallback_timeout_s=3600.0, lanes=extra)  # the synthetic ring carries no timestamps
            for i in range(N_IN):
                r.register_input(f"input_{i}")
            if ANIMATED:
                r.register_shader("soften")
            if font_book() is not None:
                r.set_fontbook(font_book())  # the renderer lays out, rasterises and draws its Text nodes itself (once per update_scene)
            for node in r.update_scene("out", OUT_W, OUT_H, scene_json()):
                if node.kind == _ffi.NODE_TEXT and font_book() is None:
                    r.set_text("out", node.index, glyphs, atlas)
            return r
        r_pipe, r_one = make_renderer(lanes[0], lanes[1:]), make_renderer(lanes[0])
        renderers = [r_pipe, r_one]
        frame_set_ring = [r_one.make_frame_set({f"input_{i}": row[i] for i in range(N_IN)}) for row in ring]
        FRAME_NS = 1_000_000_000 // 60

        tick = {id(r_pipe): 0, id(r_one): 0}  # frames rendered so far per renderer: presentation timestamps never go backwards
        #   (the warm-up, timed, serial and profiling loops all restart `step` at 0; transitions are defined on a monotonic pts)

        def step_fn(step, lane=None):
            r = r_pipe if lane is None else r_one
            t = tick[id(r)]
            tick[id(r)] += 1
            if ANIMATED and t % UPDATE_EVERY == 0 and t:
                # update_scene (the cold path of the reference, instance.rs:295-332) is part of this workload: the grid's
                # children swap places and animate for 500 ms of every 750 ms
                r.update_scene("out", OUT_W, OUT_H, scene_json(t // UPDATE_EVERY))
            r.render_packed(t * FRAME_NS, frame_set_ring[step % RING])
    else:
        FRAME_NS = 1_000_000_000 // 60
        tick = [0]
        if ANIMATED:
            # every rank evaluates the same scene at the same pts (C++ scene engine, host only): no geometry travels.  Child k of
            # the grid shows input (k + rotate) % N after an update, so the source slot -> input map changes with it.  The blur
            # layer (input 0 lives on the root) is rendered on the root: its layout node, then the shader, one surface per parity.
            from smelter_amd.scene import Scene
            engine = Scene()
            engine.update(scene_json(0), OUT_W, OUT_H)
            inner_s = [ctx.surface(LAYER_W, LAYER_H) for _ in range(2)] if rank == 0 else None
            layer_s = [ctx.surface(LAYER_W, LAYER_H) for _ in range(2)] if rank == 0 else None
            label = layer_s[0] if rank == 0 else None
        # the exchange goes through the C ABI (smr_comm_create_rank + smr_gather_tiles: RCCL send / recv on the ctx stream);
        # torch.distributed only carries the 128-byte communicator id and the barriers
        comm = None
        watchdog = Watchdog(args.watchdog_seconds, rank, world, args.config)
        watchdog.beat("communicator set-up (smr_comm_create_rank: ncclCommInitRank)")
        if world > 1:
            uid = torch.zeros(hip.COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid = torch.tensor(list(hip.Comm.unique_id()), dtype=torch.uint8, device="cuda")
            dist.broadcast(uid, src=0)
            try:
                comm = hip.Comm.rank(ctx, world, rank, bytes(uid.cpu().tolist()))
                ok = 1
            except Exception as e:  # (e.g. librccl not loadable on this node): the torch.distributed send / recv twin of the same exchange
                print(f"[bench] rank {rank}: smr_comm_create_rank failed ({e}); falling back to torch.distributed point-to-point", file=sys.stderr)
                comm, ok = None, 0
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # all ranks use the same transport
            if int(flag.item()) == 0 and comm is not None:
                comm.close()
                comm = None
        sharded = smr_dist.ShardedCompositor(ctx, hip, plan, rank, layouts, res, input_source_slot, label, torch, dist, comm=comm)

        def step_fn(step):
            watchdog.beat(f"step {step}")
            t = tick[0]
            tick[0] += 1
            if ANIMATED:
                rot = t // UPDATE_EVERY
                if t % UPDATE_EVERY == 0 and t:
                    engine.update(scene_json(rot), OUT_W, OUT_H)
                slots = [(i - rot) % N_IN for i in range(N_IN)]  # input i is child (i - rotate) % N of the grid
                sharded.set_layouts(engine.layouts(0, t * FRAME_NS, res), slots)
                if rank == 0:
                    par = t & 1
                    ctx.render_layouts(INNER_LAYOUTS, [ring[step % RING][0]], LAYER_W, LAYER_H, out_rgba=inner_s[par])
                    ctx.gaussian_blur(inner_s[par], LAYER_SIGMA, dst=layer_s[par])
                    sharded.label = layer_s[par]
            # frame k's tiles travel over xGMI while frame k-1 is composed (two tile sets, two output frames)
            sharded.step_pipelined(ring[step % RING], out_for(step) if rank == 0 else None)

    def barrier():
        if not single:
            watchdog.beat("barrier: flush + sync")
            sharded.flush()  # the frame still in flight
        if world > 1:
            dist.barrier()
        for c in lanes:
            c.sync()
        torch.cuda.synchronize()

    # one-time set-up, not a step: every lane renders two frames so that its weight bands, tile classes, scratch surfaces and kernel
    # attributes exist before the W warm-up steps and the K timed ones (a context's first frame costs ~1 ms; `config.prime_frames`)
    PRIME = 2 * max(1, n_lanes)
    for s in range(PRIME):
        step_fn(s)
    barrier()
    # ... and the device is brought to the state a running compositor is in: untimed frames for `--prime-seconds` (default 0.25 s) so that the
    # K timed steps are not also a measurement of the clock ramp out of idle (a 20-step run lasts a millisecond; the same loop over 12 s is
    # `value_long`).  Reported as config.prime_seconds / prime_frames; 0 restores the bare W warm-up steps.
    t_prime = time.perf_counter()
    while time.perf_counter() - t_prime < args.prime_seconds:
        for s in range(32):
            step_fn(PRIME + s)
        PRIME += 32
        barrier()
    for s in range(args.warmup):
        step_fn(s)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step_fn(s)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the same loop over >= 2 s of device work (a K = 20 run lasts about a millisecond: the pipeline's fill and drain weigh on it and an outside
    # observer sampling GPU activity cannot see it), reported as `value_long` beside `value` — never instead of it
    long_steps = args.steps if args.no_long else min(max(args.steps, int(args.long_seconds * args.steps / max(elapsed, 1e-9))), 2000000)
    tl = time.perf_counter()
    for s in range(long_steps):
        step_fn(s)
    barrier()
    long_elapsed = time.perf_counter() - tl
    if world > 1:
        t = torch.tensor([long_elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        long_elapsed = float(t.item())
    serial_fps = None
    if single and n_lanes > 1:
        # the same K steps on ONE renderer (frames strictly one after the other on one stream), reported beside `value`
        t1 = time.perf_counter()
        for s in range(args.steps):
            step_fn(s, ctx)
        barrier()
        serial_fps = args.steps / (time.perf_counter() - t1)

    result = None
    if rank == 0:
        fps = args.steps / elapsed
        result = {
            "metric": {1: "composited frames/sec, 4x1080p->1 1080p scene", 2: "composited frames/sec, 8x1080p->1 4K scene",
                       3: "composited frames/sec, 8x4K->1 4K scene", 4: "composited frames/sec, 16x1080p animated grid + blur layer->1 4K scene"}[args.config],
            "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 5),
            "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
            "value_long": {"frames_per_s": round(long_steps / long_elapsed, 2), "steps": long_steps, "seconds": round(long_elapsed, 3),
                           "what": f"the same timed loop over ~{args.long_seconds:g} s of device work, run before any CPU work (fill / drain of the pipeline amortised; "
                                   "long enough for an outside GPU-activity sampler to see)"},
            "dtype": "u8 (f32 colour conversion: the WGSL sequence value for value; Lanczos on f16-pair MFMA with f32 accumulate, f16 resampler intermediate)"
            if args.ingest != "valu" else "u8 (f32 arithmetic, f16 resampler intermediate)", "data": "synthetic",
            "config": {"workload": {1: "configs[1]: 4x1080p YUV420 inputs -> 1920x1080 YUV420, Tiles, rescale + blend only, GpuOptimized",
                                    2: "configs[2]: 8x1080p YUV420 inputs tiled -> 3840x2160 YUV420, Tiles + Rescaler(border_radius 24) "
                                       "+ text label per tile, GpuOptimized (linear-light Lanczos3 + blend)",
                                    3: "configs[3] on ONE GPU: 8x4K YUV420 inputs tiled -> 3840x2160 YUV420, same scene as configs[2]",
                                    4: "configs[4] on ONE GPU: 16x1080p YUV420 inputs in an animated Tiles grid (scene update every 45 frames, "
                                       "500 ms cubic-bezier transitions) + one 960x540 layer through the gaussian-blur shader -> 3840x2160 YUV420"}[args.config],
                       "inputs": N_IN, "input_resolution": [IN_W, IN_H], "output_resolution": [OUT_W, OUT_H],
                       "layouts": len(layouts), "prime_frames": PRIME, "prime_seconds": args.prime_seconds, "input_ring": RING, "frames_in_flight": len(lanes),
                       "frames_per_s_one_in_flight": round(serial_fps, 2) if serial_fps else None,
                       "per_frame_host_work": "smr_renderer_render: frame set -> layout maths at pts (C++ scene engine) -> parameter pack -> 3 kernels (convert, resample, compose)"
                       if single else ("layout maths at pts on every rank (C++ scene engine) -> ingest per shard -> gather -> blur layer + compose on the root"
                                       if ANIMATED else "pre-flattened layout list (scene engine, once) -> ingest per shard -> gather -> compose"),
                       "parallelism": "single GPU" if single else f"inputs sharded over {world} GPUs, RCCL gather to rank 0"},
            "frame": {"algorithmic_bytes": ALGO_BYTES_PER_FRAME, "achieved_GBps": round(ALGO_BYTES_PER_FRAME * fps / 1e9, 2),
                      "frac_of_hbm_peak": round(ALGO_BYTES_PER_FRAME * fps / 1e9 / HBM_PEAK_GBPS, 5)},
        }

    # ---- per-kernel timing with HIP events on the ctx stream (outside the timed region: per-launch events
    #      serialise the pipeline), then the dominant kernel's roofline entry
    if single:
        ctx.profile_reset()
        ctx.profile_enable(True)
        for s in range(min(args.steps, 200)):
            step_fn(s, ctx)
        ctx.sync()
        prof = ctx.profile_read()
        ctx.profile_enable(False)
        stages = {k: {"avg_us": round(1000.0 * ms / n, 3), "launches": n} for k, (ms, n) in prof.items() if n}
        tile_bytes = 0
        for L in layouts:
            if L.type == 0 and res[L.source_index] == (IN_W, IN_H):
                tile_bytes += max(int(np.floor(L.width + 0.5)), 1) * max(int(np.floor(L.height + 0.5)), 1) * 4
        # the node textures of the default route (written by the converter, read by the resampler): RGB12 — 3 bytes per pixel — where the
        # resampler runs a class build (configs[2], configs[3]: SMR_OPT_COMPACT_NODES), RGBA8 elsewhere
        node_bytes = N_IN * IN_W * IN_H * (3 if args.config in (2, 3) else 4) if "ingest" in stages else 0
        kernel_bytes = {
            "ingest": N_IN * yuv420_bytes(IN_W, IN_H) + node_bytes,                                     # converter: planes in, node textures out
            "fused_ingest_resample": (node_bytes or N_IN * yuv420_bytes(IN_W, IN_H)) + tile_bytes,       # resampler: nodes (or planes) in, tiles out
            "fused_compose_output": tile_bytes + yuv420_bytes(OUT_W, OUT_H),                            # reads the tiles once, writes Y, U, V once
        }
        knames = {"ingest": "k_yuv420_to_rgba" if args.convert == "auto" else {"general": "k_yuv_to_rgba", "block4x2": "k_yuv_to_rgba_batch"}[args.convert],
                  "fused_ingest_resample": "k_ingest_resample" if args.ingest == "valu" else "k_ingest_wave", "fused_compose_output": "k_compose_output"}
        result["roofline"] = roofline_block(args, stages, kernel_bytes, knames, result["ms_per_step"])
        result["kernels"] = stages
        # latency: one frame in flight, inputs resident -> output planes resident in HBM
        lat, enq = [], []
        for s in range(args.latency_frames):
            t1 = time.perf_counter()
            step_fn(s, ctx)
            t2 = time.perf_counter()
            ctx.sync()
            lat.append(time.perf_counter() - t1)
            enq.append(t2 - t1)
        lat, enq = np.array(lat) * 1e3, np.array(enq) * 1e3
        result["latency_ms"] = {"p50": round(float(np.percentile(lat, 50)), 4), "p99": round(float(np.percentile(lat, 99)), 4),
                                "frames": args.latency_frames, "definition": "host enqueue -> output planes resident in HBM, 1 frame in flight",
                                "host_call_p50": round(float(np.percentile(enq, 50)), 4),
                                "host_call": "the smr_renderer_render call alone (layout maths, parameter pack, two launches) on an idle device"}
        # ... and to host-visible: the same plus the stream-ordered read-back of the output planes into pinned host memory
        from smelter_amd.renderer import BorrowedFrame
        lat, host_out = [], None
        for s in range(min(args.latency_frames, 300)):
            t1 = time.perf_counter()
            step_fn(s, ctx)
            of = r_one.output(0)
            if host_out is None:
                host_out = of.pinned_planes()
            of.download_async(host_out)
            ctx.sync()
            lat.append(time.perf_counter() - t1)
        lat = np.array(lat) * 1e3
        result["latency_host_visible_ms"] = {"p50": round(float(np.percentile(lat, 50)), 4), "p99": round(float(np.percentile(lat, 99)), 4),
                                             "frames": len(lat), "definition": "host enqueue -> output planes in pinned host memory "
                                             f"({yuv420_bytes(OUT_W, OUT_H)} B over PCIe), 1 frame in flight"}
        # the HBM denominator as measured on this device (SURVEY.md §8d asks for spec and measured): a 1 GiB device-to-device copy
        a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
        if "roofline" in result:
            result["roofline"]["peak_measured_copy_GBps"] = round(copy_gbps, 1)
            result["roofline"]["frac_of_measured_copy"] = round(result["roofline"]["achieved"] / copy_gbps, 5)
        if args.transfers:
            # informational (never `value`): the same frames handed over as host buffers and read back to the host —
            # smr_frame_upload of every input plane + render + smr_frame_download of the output planes, one frame at a time
            from smelter_amd import synth
            host_planes = [synth.test_input(i, IN_W, IN_H, noise_seed=99 + i) for i in range(N_IN)]
            row0 = ring[0]
            t1 = time.perf_counter()
            reps = 60
            for s in range(reps):
                for i in range(N_IN):
                    row0[i].upload(host_planes[i])
                n_out = r_one.render_packed((tick[id(r_one)] + s) * FRAME_NS, frame_set_ring[0])
                assert n_out == 1
                from smelter_amd.renderer import BorrowedFrame
                r_one.output(0).download()
            dt = (time.perf_counter() - t1) / reps
            moved = N_IN * yuv420_bytes(IN_W, IN_H) + yuv420_bytes(OUT_W, OUT_H)
            result["pcie_inclusive"] = {"frames_per_s": round(1.0 / dt, 1), "ms_per_frame": round(dt * 1e3, 3),
                                        "host_bytes_per_frame": moved, "GBps": round(moved / dt / 1e9, 2),
                                        "note": "pageable host buffers, blocking copies, 1 frame in flight"}
            # the same with pinned host buffers and stream-ordered copies (smr_host_alloc / smr_frame_*_async), two frames in
            # flight on two renderers: frame k+1's upload overlaps frame k's kernels and read-back
            k_lanes = min(2, len(lanes))
            pin_in = [[lanes[k].frame(hip.FRAME_PLANAR_YUV420, IN_W, IN_H) for _ in range(N_IN)] for k in range(k_lanes)]
            pin_host = [[f.pinned_planes() for f in pin_in[k]] for k in range(k_lanes)]
            for k in range(k_lanes):
                for i in range(N_IN):
                    for dst, src in zip(pin_host[k][i], host_planes[i]):
                        dst[...] = src
            rt = [make_renderer(lanes[k]) for k in range(k_lanes)]  # one single-lane renderer per frame in flight (uploads ride on its stream)
            renderers += rt
            pin_sets = [rt[k].make_frame_set({f"input_{i}": pin_in[k][i] for i in range(N_IN)}) for k in range(k_lanes)]
            out_host = [None] * k_lanes
            t1 = time.perf_counter()
            reps = 200
            for s in range(reps):
                k = s % k_lanes
                lanes[k].sync()  # the lane's previous frame (its host buffers are free again)
                for i in range(N_IN):
                    pin_in[k][i].upload_async(pin_host[k][i])
                rt[k].render_packed(s * FRAME_NS, pin_sets[k])
                of = rt[k].output(0)
                if out_host[k] is None:
                    out_host[k] = of.pinned_planes()
                of.download_async(out_host[k])
            for k in range(k_lanes):
                lanes[k].sync()
            dt = (time.perf_counter() - t1) / reps
            result["pcie_inclusive_pinned"] = {"frames_per_s": round(1.0 / dt, 1), "ms_per_frame": round(dt * 1e3, 3),
                                               "GBps": round(moved / dt / 1e9, 2),
                                               "note": f"pinned host buffers, stream-ordered copies, {k_lanes} frames in flight"}
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(layouts, res)
        if args.config == 2 and not args.no_target:
            # BASELINE.json north_star's target configuration — 8x4K30 inputs -> one 4K output on a SINGLE MI355X — measured by the same
            # program (child process, same library) so that the judged line carries it: frames/s, roofline, latency over >= 2000 frames
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--config", "3", "--steps", "300", "--warmup", "30", "--no-cpu-baseline",
                   "--latency-frames", "2000", "--ingest", args.ingest, "--long-seconds", "2"]
            try:
                child = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                tj = json.loads([ln for ln in child.stdout.splitlines() if ln.startswith("{")][-1])
                result["target"] = {"workload": tj["config"]["workload"], "frames_per_s": tj["value"], "ms_per_frame": tj["ms_per_step"],
                                    "frames_per_s_long": tj.get("value_long", {}).get("frames_per_s"),
                                    "frames_per_s_one_in_flight": tj["config"]["frames_per_s_one_in_flight"], "goal_frames_per_s": 60,
                                    "frame": tj["frame"], "roofline": tj.get("roofline"), "kernels": tj.get("kernels"),
                                    "latency_ms": tj.get("latency_ms"), "latency_host_visible_ms": tj.get("latency_host_visible_ms")}
            except Exception as e:  # the judged line must survive a failing child
                result["target"] = {"error": f"{type(e).__name__}: {e}"}
            # the N > 1 code path (ingest per shard -> gather -> compose on the root, what `--gpus N` runs) with ONE rank on the same workload: the N = 1
            # point a SCALE record's curve starts from, measured by the same program beside the renderer path's figure above
            cmd = [sys.executable, os.path.abspath(__file__), "--force-sharded", "--steps", "300", "--warmup", "30", "--no-cpu-baseline", "--no-long",
                   "--ingest", args.ingest]
            try:
                child = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                tj = json.loads([ln for ln in child.stdout.splitlines() if ln.startswith("{")][-1])
                result["sharded_one_rank"] = {"workload": tj["config"]["workload"], "frames_per_s": tj["value"], "ms_per_frame": tj["ms_per_step"],
                                              "kernels": tj.get("kernels"), "roofline": tj.get("roofline"),
                                              "what": "bench.py --force-sharded: configs[3] through ShardedCompositor with world_size 1 (no exchange): SCALE's N = 1 point"}
            except Exception as e:
                result["sharded_one_rank"] = {"error": f"{type(e).__name__}: {e}"}
    else:
        # N > 1: every rank runs a few more steps with per-launch HIP events on; rank 0 reports the kernels of the root
        # (its own shard's ingest + the compose of the gathered tiles) and the dominant one's roofline entry
        ctx.profile_reset()
        ctx.profile_enable(True)
        for s in range(40):
            step_fn(s)
        barrier()
        prof = ctx.profile_read()
        ctx.profile_enable(False)
        if rank == 0:
            stages = {k: {"avg_us": round(1000.0 * ms / n, 3), "launches": n} for k, (ms, n) in prof.items() if n}
            tile_px = {}
            for L in layouts:
                if L.type == 0 and res[L.source_index] == (IN_W, IN_H):
                    tile_px[L.source_index] = max(int(np.floor(L.width + 0.5)), 1) * max(int(np.floor(L.height + 0.5)), 1) * 4
            tile_bytes = sum(tile_px.values())
            local_tiles = sum(tile_px[input_source_slot[i]] for i in my_inputs if input_source_slot[i] in tile_px)
            # (node textures of the batch path: RGB12 — 3 bytes per pixel — for the class builds of configs[2] / configs[3], RGBA8 elsewhere)
            node_b = len(my_inputs) * IN_W * IN_H * (3 if args.config in (2, 3) else 4) if "ingest" in stages else 0
            kernel_bytes = {"ingest": len(my_inputs) * yuv420_bytes(IN_W, IN_H) + node_b,
                            "fused_ingest_resample": (node_b or len(my_inputs) * yuv420_bytes(IN_W, IN_H)) + local_tiles,
                            "fused_compose_output": tile_bytes + yuv420_bytes(OUT_W, OUT_H)}
            knames = {"ingest": "k_yuv420_to_rgba", "fused_ingest_resample": "k_ingest_wave", "fused_compose_output": "k_compose_output"}
            # the single-GPU definition: the algorithmic bytes this rank's launches stand for (its shard's inputs in their native format; on the root
            # also the output frame) / the sum of its launches' mean durations
            rb = roofline_block(args, {k: v for k, v in stages.items() if k in kernel_bytes}, kernel_bytes, knames, result["ms_per_step"],
                                algo_bytes=len(my_inputs) * yuv420_bytes(IN_W, IN_H) + yuv420_bytes(OUT_W, OUT_H))
            if rb:
                rb["rank"] = 0
                rb["traffic"], rb["traffic_per_kernel"], rb["traffic_over_algorithmic"], rb["traffic_stale"] = None, None, None, False
                rb["traffic_source"] = "not collected for the sharded path"
                rb.pop("issue", None)
                rb["definition"] += " — rank 0's share of the frame: its shard's input bytes + the output frame"
                result["roofline"] = rb
            result["kernels"] = stages
            # the exchange step against the xGMI point-to-point roof: every peer sends its tiles to the root over its own link
            per_peer = {}
            for i in plan.remote_inputs():
                per_peer[plan.owner(i)] = per_peer.get(plan.owner(i), 0) + tile_px.get(input_source_slot[i], 0)
            if per_peer:
                fps = result["value"]
                worst = max(per_peer.values())
                result["exchange"] = {"bytes_per_frame": sum(per_peer.values()), "peers": len(per_peer), "max_bytes_per_link": worst,
                                      "link_GBps_achieved": round(worst * fps / 1e9, 3), "link_peak_GBps": 153.0,
                                      "frac_of_link_peak": round(worst * fps / 1e9 / 153.0, 5),
                                      "transport": "smr_gather_tiles (C ABI, RCCL send / recv on the ctx stream)" if comm is not None
                                      else "torch.distributed isend / irecv (fallback)",
                                      "note": "dst-sized RGBA8 tiles gathered on the root, one xGMI link per peer"}

    if not single:
        watchdog.stop()
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if single:
        for r in renderers:
            r.close()
    for c in lanes[1:]:
        c.close()
    ctx.close()


if __name__ == "__main__":
    main()
