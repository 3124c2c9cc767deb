#!/usr/bin/env python3
"""Builds integration/smelter-render-hip.patch: the Rust seam that routes smelter-render through libsmr_hip.

  python tools/make_rust_patch.py [--reference /root/reference]

* src/hip/sys.rs is GENERATED from include/smr.h (every SMR_API entry point, same order, C types mapped to Rust), so the binding
  cannot drift from the header; tests/test_rust_patch.py re-parses both and compares names and arities.
* the adapter (src/hip/mod.rs: FFI owners), the seam (src/state/hip.rs: the render loop on libsmr_hip; src/transformations/layout/hip.rs:
  LayoutNode::render on libsmr_hip, inside the module whose private items it needs) and the hunks against state.rs, state/node.rs,
  state/frame_pre_processor.rs, transformations/layout.rs, transformations/shader.rs, transformations/shader/node.rs, error.rs and
  lib.rs are written here and diffed against the reference checkout, so the patch applies with `git apply` at the surveyed revision.
* No enum gains a variant and no public struct a field (RenderingMode is matched exhaustively in ~20 places and RendererOptions is built
  by three other crates): the seam is switched on by SMELTER_HIP_DEVICE=<ordinal> in the environment of the process.
The patch cannot be compiled in this environment (no cargo / rustc); it is reviewed text, kept honest by tests/test_rust_patch.py:
the binding against the header, and every crate item the added code names against the reference's definitions and visibility.
"""
from __future__ import annotations

import argparse
import difflib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BASE = {"uint32_t": "u32", "int32_t": "i32", "int": "c_int", "int64_t": "i64", "uint8_t": "u8", "size_t": "usize", "float": "f32",
        "double": "f64", "void": "c_void", "char": "c_char", "uint64_t": "u64"}


def c_decls(header: str):
    """[(name, ret_c, [param_c, ...])] for every SMR_API declaration, in header order."""
    text = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    out = []
    for m in re.finditer(r"SMR_API\s+([^;(]+?)\b(smr_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        plist = [] if params in ("void", "") else [p.strip() for p in params.split(",")]
        out.append((name, ret, plist))
    return out


def rust_type(decl: str, is_param=True):
    """C declarator (with or without a name) -> (rust type, name)."""
    arr = re.search(r"\[(\w*)\]\s*$", decl)
    if arr:
        decl = decl[: arr.start()].strip()
    toks = re.findall(r"\*|\w+", decl)
    name = None
    if is_param and toks and toks[-1] not in ("const", "*") and len([t for t in toks if t not in ("const", "*", "struct", "unsigned")]) > 1:
        name = toks.pop()
    i, base_const, base = 0, False, None
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            base_const = True
        elif toks[i] not in ("struct",):
            base = toks[i]
        i += 1
    t = BASE.get(base, base)
    const = base_const
    while i < len(toks):
        assert toks[i] == "*", decl
        t = ("*const " if const else "*mut ") + t
        const = i + 1 < len(toks) and toks[i + 1] == "const"
        i += 2 if const else 1
    if arr:  # T name[N] as a parameter is T *name
        t = ("*const " if const else "*mut ") + t
    return t, name


def c_enums(header: str):
    """[(enum name, [(constant, value)])] for every `typedef enum` of the header."""
    text = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    out = []
    for m in re.finditer(r"typedef\s+enum\s+(\w+)\s*\{(.*?)\}", text, flags=re.S):
        consts, nxt = [], 0
        for item in m.group(2).split(","):
            item = item.strip()
            if not item:
                continue
            name, _, val = item.partition("=")
            nxt = int(val.strip(), 0) if val.strip() else nxt
            consts.append((name.strip(), nxt))
            nxt += 1
        out.append((m.group(1), consts))
    return out


def gen_sys_rs(header: str) -> str:
    decls = c_decls(header)
    enums = []
    for ename, consts in c_enums(header):
        ty = "c_int" if ename == "smr_status" else "u32"  # status codes are returned as int (negative); the rest are uint32_t arguments
        enums.append(f"// enum {ename}")
        enums += [f"pub const {n}: {ty} = {v};" for n, v in consts]
    lines = []
    for name, ret, params in decls:
        rparams = []
        for k, p in enumerate(params):
            t, n = rust_type(p)
            n = n or f"arg{k}"
            if n in ("in", "type", "ref", "fn", "box", "move", "loop", "match", "mod", "self", "where"):
                n += "_"
            rparams.append(f"{n}: {t}")
        rt, _ = rust_type(ret, is_param=False)
        tail = "" if rt == "c_void" else f" -> {rt}"
        lines.append(f"    pub fn {name}({', '.join(rparams)}){tail};")
    abi = re.search(r"^#define SMR_ABI_VERSION (\d+)", header, re.M).group(1)
    head = SYS_HEAD.replace("@@ENUMS@@", "\n".join(enums)).replace("@@ABI@@", abi)
    return head + "\n".join(lines) + "\n}\n"


SYS_HEAD = '''//! Raw binding of libsmr_hip (include/smr.h) — GENERATED by tools/make_rust_patch.py of the smelter_amd repository from that
//! header: one `pub fn` per SMR_API entry point, in header order.  Do not edit by hand; regenerate when the header changes.
#![allow(non_camel_case_types, dead_code, clippy::too_many_arguments)]
use std::os::raw::{c_char, c_int, c_void};

// opaque handles
#[repr(C)]
pub struct smr_ctx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct smr_surface {
    _p: [u8; 0],
}
#[repr(C)]
pub struct smr_scene {
    _p: [u8; 0],
}
#[repr(C)]
pub struct smr_renderer {
    _p: [u8; 0],
}
#[repr(C)]
pub struct smr_comm {
    _p: [u8; 0],
}
#[repr(C)]
pub struct smr_fontbook {
    _p: [u8; 0],
}

@@ENUMS@@
pub const SMR_MAX_MASKS: usize = 20; // MAX_MASKS_COUNT (transformations/layout/params.rs:15)
pub const SMR_NO_SOURCE: u32 = 0xffff_ffff;
pub const SMR_COMM_ID_BYTES: usize = 128;
pub const SMR_ABI_VERSION: u32 = @@ABI@@; // the header this file was generated from; HipCtx::new compares it with smr_abi_version()

#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_frame { pub format: u32, pub width: u32, pub height: u32, pub planes: [*mut smr_surface; 3] }
#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_surface_info { pub width: u32, pub height: u32, pub format: u32, pub owned: u32, pub pitch: usize, pub dptr: *mut c_void }
#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_mask { pub radius: [f32; 4], pub top: f32, pub left: f32, pub width: f32, pub height: f32 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_layout {
    pub top: f32, pub left: f32, pub width: f32, pub height: f32, pub rotation_degrees: f32,
    pub border_radius: [f32; 4],                      // tl, tr, br, bl (params.rs:337-344)
    pub type_: u32,                                    // 0 ChildNode, 1 Color, 2 BoxShadow (params.rs:223-303)
    pub source_index: u32,
    pub color: [f32; 4], pub border_color: [f32; 4],  // convert_to_shader_color (wgpu/utils.rs:51-81), premultiplied
    pub border_width: f32,
    pub crop: [f32; 4],                               // top, left, width, height (layout.rs:39-45)
    pub blur_radius: f32,
    pub masks_len: u32,
    pub masks: [smr_mask; SMR_MAX_MASKS],
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_resample_plan {
    pub kind: c_int, pub levels: [c_int; 2], pub reduced_w: c_int, pub reduced_h: c_int, pub axis: [c_int; 2], pub scale: [f32; 2],
    pub offset: [f32; 2], pub perp_offset: [c_int; 2], pub mid_w: c_int, pub mid_h: c_int,
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_glyph { pub dst_x: i32, pub dst_y: i32, pub w: i32, pub h: i32, pub atlas_x: i32, pub atlas_y: i32, pub color: [f32; 4] }
#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_source { pub kind: u32, pub surface: *const smr_surface, pub frame: *const smr_frame }
#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_circle_layout { pub left_px: u32, pub top_px: u32, pub width_px: u32, pub height_px: u32, pub background_color: [f32; 4] }
#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_gaussian_blur_params { pub sigma: f32 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_text_params {
    pub text: *const c_char, pub font_family: *const c_char, pub style: *const c_char, pub weight: *const c_char, pub wrap: *const c_char,
    pub align: *const c_char, pub font_size: f32, pub line_height: f32, pub max_width: f32, pub max_height: f32,
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_text_run { pub glyphs: *const smr_glyph, pub n_glyphs: u32, pub atlas: *const u8, pub atlas_w: u32, pub atlas_h: u32 }
pub type smr_text_measure_fn =
    Option<unsafe extern "C" fn(user: *mut c_void, params: *const smr_text_params, widest_line: *mut f32, line_count: *mut u32) -> c_int>;
#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_scene_node {
    pub kind: u32, pub parent: i32, pub n_children: u32, pub width: u32, pub height: u32,
    pub ref_id: *const c_char, pub id: *const c_char, pub payload: *const c_char,
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_input_frame { pub input_id: *const c_char, pub frame: *const smr_frame, pub pts_ns: i64 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct smr_output_frame { pub output_id: *const c_char, pub frame: smr_frame, pub ctx: *mut smr_ctx }

#[link(name = "smr_hip")]
unsafe extern "C" {
'''

MOD_RS = r"""//! smelter-render's per-frame rasteriser on libsmr_hip (MI355X, HIP) instead of wgpu — switched on by SMELTER_HIP_DEVICE=<ordinal>.
//!
//! The scene maths stays where it is (scene/*, transformations/layout/flatten.rs): LayoutProvider::layouts + flatten produce
//! the RenderLayout list exactly as for wgpu; only what happens below that list changes:
//!   InputTexture::upload + convert_to_node_texture   -> smr_frame_upload (conversion is fused into the layout call)
//!   LayoutNode::render (resample + LayoutShader)      -> smr_render_layouts        (transformations/layout/hip.rs)
//!   read_outputs (rgba_to_yuv + padded copy + map)    -> fused into the root's smr_render_layouts + smr_frame_download
//!   ShaderNode::render of the in-tree WGSL shaders     -> smr_builtin_shader        (matched by source text, see builtin_shader_id)
//!   FramePreProcessor::process_to_bytes               -> smr_frame_preprocess
//! Text and image nodes keep rendering through wgpu (glyphon, the decoded assets) into their NodeTexture; its pixels are read back
//! and handed over with smr_surface_upload — text once per node, images every frame (they may be animated).  Web views and
//! user-written WGSL shaders need the wgpu renderer: a scene that uses them fails with a validation error in this mode.
//! This file owns the FFI objects; the render loop is src/state/hip.rs.
pub mod sys;

use std::{ffi::CStr, ptr, sync::Arc};

use bytes::Bytes;
use sys::*;

use crate::{
    Frame, FrameData, OutputFrameFormat, RenderingMode, Resolution, YuvPlanes,
    wgpu::{WgpuCtx, WgpuError, texture::TextureExt},
};

/// Owner of one `smr_ctx` (device, stream, caches).  `Renderer` serialises every call behind its mutex (state.rs:55).
pub(crate) struct HipCtx {
    raw: *mut smr_ctx,
    /// colour handling of the layout pass: CpuOptimized blends in sRGB space, everything else in linear light
    pub(crate) mode: RenderingMode,
    pub(crate) max_layouts: usize,
}
unsafe impl Send for HipCtx {}
unsafe impl Sync for HipCtx {}

impl HipCtx {
    pub(crate) fn new(hip_device: i32, mode: RenderingMode, max_layouts: usize) -> Result<Arc<Self>, WgpuError> {
        let smr_mode = match mode {
            RenderingMode::CpuOptimized => SMR_MODE_CPU_OPTIMIZED,
            RenderingMode::GpuOptimized | RenderingMode::WebGl => SMR_MODE_GPU_OPTIMIZED,
        };
        let abi = unsafe { smr_abi_version() };
        if abi != SMR_ABI_VERSION {
            return Err(WgpuError::Internal(format!("libsmr_hip has ABI version {abi}, this binding was generated for {SMR_ABI_VERSION}")));
        }
        let mut raw = ptr::null_mut();
        let rc = unsafe { smr_ctx_create(hip_device, smr_mode, max_layouts as u32, ptr::null_mut(), &mut raw) };
        if rc != SMR_OK {
            return Err(WgpuError::Internal(format!("smr_ctx_create(device {hip_device}) failed: {rc}")));
        }
        Ok(Arc::new(Self { raw, mode, max_layouts }))
    }
    /// SMELTER_HIP_DEVICE=<ordinal> in the environment switches the seam on.
    pub(crate) fn device_from_env() -> Option<i32> {
        std::env::var("SMELTER_HIP_DEVICE").ok().and_then(|v| v.parse::<i32>().ok())
    }
    pub(crate) fn raw(&self) -> *mut smr_ctx {
        self.raw
    }
    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(smr_last_error(self.raw)) }.to_string_lossy().into_owned()
    }
    /// Return codes -> the error classes WgpuErrorScope::pop reports today (wgpu.rs:78-97).
    pub(crate) fn check(&self, rc: i32) -> Result<i32, WgpuError> {
        match rc {
            0.. => Ok(rc),
            SMR_ERR_INVALID => Err(WgpuError::Validation(self.last_error())),
            SMR_ERR_OOM => Err(WgpuError::OutOfMemory(self.last_error())),
            _ => Err(WgpuError::Internal(self.last_error())),
        }
    }
}
impl Drop for HipCtx {
    fn drop(&mut self) {
        unsafe { smr_ctx_destroy(self.raw) }
    }
}

/// InputTexture (state/input_texture.rs:69-201): one device frame per input, re-created when format or resolution change.
pub(crate) struct HipInput {
    ctx: Arc<HipCtx>,
    frame: Option<smr_frame>,
}
unsafe impl Send for HipInput {}

impl HipInput {
    pub(crate) fn new(ctx: Arc<HipCtx>) -> Self {
        Self { ctx, frame: None }
    }
    pub(crate) fn clear(&mut self) {
        if let Some(mut f) = self.frame.take() {
            unsafe { smr_frame_destroy(self.ctx.raw, &mut f) }
        }
    }
    pub(crate) fn frame(&self) -> Option<&smr_frame> {
        self.frame.as_ref()
    }
    pub(crate) fn upload(&mut self, frame: Frame) -> Result<(), WgpuError> {
        let (format, planes): (u32, Vec<&Bytes>) = match &frame.data {
            FrameData::PlanarYuv420(p) => (SMR_FRAME_PLANAR_YUV420, vec![&p.y_plane, &p.u_plane, &p.v_plane]),
            FrameData::PlanarYuv422(p) => (SMR_FRAME_PLANAR_YUV422, vec![&p.y_plane, &p.u_plane, &p.v_plane]),
            FrameData::PlanarYuv444(p) => (SMR_FRAME_PLANAR_YUV444, vec![&p.y_plane, &p.u_plane, &p.v_plane]),
            FrameData::PlanarYuvJ420(p) => (SMR_FRAME_PLANAR_YUVJ420, vec![&p.y_plane, &p.u_plane, &p.v_plane]),
            FrameData::Nv12(p) => (SMR_FRAME_NV12, vec![&p.y_plane, &p.uv_planes]),
            FrameData::InterleavedUyvy422(b) => (SMR_FRAME_UYVY422, vec![b]),
            FrameData::InterleavedYuyv422(b) => (SMR_FRAME_YUYV422, vec![b]),
            FrameData::Bgra(b) => (SMR_FRAME_BGRA, vec![b]),
            FrameData::Argb(b) => (SMR_FRAME_ARGB, vec![b]),
            FrameData::Rgba8UnormWgpuTexture(_) | FrameData::Nv12WgpuTexture(_) => {
                return Err(WgpuError::Validation("wgpu textures cannot be inputs of the HIP rasteriser".into()));
            }
        };
        let (w, h) = (frame.resolution.width as u32, frame.resolution.height as u32);
        let same = self.frame.as_ref().map(|f| (f.format, f.width, f.height)) == Some((format, w, h));
        if !same {
            self.clear();
            let mut f = smr_frame { format, width: w, height: h, planes: [ptr::null_mut(); 3] };
            self.ctx.check(unsafe { smr_frame_create(self.ctx.raw, format, w, h, &mut f) })?;
            self.frame = Some(f);
        }
        let mut host = [ptr::null::<std::os::raw::c_void>(); 3];
        for (k, p) in planes.iter().enumerate() {
            host[k] = p.as_ptr().cast();
        }
        let f = self.frame.as_ref().unwrap();
        self.ctx.check(unsafe { smr_frame_upload(self.ctx.raw, f, host.as_ptr()) }).map(|_| ())
    }
}
impl Drop for HipInput {
    fn drop(&mut self) {
        self.clear()
    }
}

/// NodeTexture (state/node_texture.rs:11-163) of text / image / shader / nested layout nodes: an RGBA8 surface, size-keyed.
pub(crate) struct HipNodeTexture {
    ctx: Arc<HipCtx>,
    surface: Option<(*mut smr_surface, Resolution)>,
    /// text nodes render once (text_renderer.rs:73-75): their pixels are handed over once
    pub(crate) filled: bool,
}
unsafe impl Send for HipNodeTexture {}

impl HipNodeTexture {
    pub(crate) fn new(ctx: Arc<HipCtx>) -> Self {
        Self { ctx, surface: None, filled: false }
    }
    pub(crate) fn surface(&self) -> Option<(*mut smr_surface, Resolution)> {
        self.surface
    }
    fn clear(&mut self) {
        if let Some((s, _)) = self.surface.take() {
            unsafe { smr_surface_destroy(self.ctx.raw, s) }
        }
        self.filled = false;
    }
    pub(crate) fn ensure_size(&mut self, res: Resolution) -> Result<*mut smr_surface, WgpuError> {
        if let Some((s, r)) = self.surface {
            if r == res {
                return Ok(s);
            }
        }
        self.clear();
        let mut s = ptr::null_mut();
        self.ctx.check(unsafe { smr_surface_create(self.ctx.raw, res.width as u32, res.height as u32, SMR_PX_RGBA8, &mut s) })?;
        self.surface = Some((s, res));
        Ok(s)
    }
    /// The pixels of a wgpu node texture (premultiplied RGBA8, the bytes wgpu stores) -> this surface.
    pub(crate) fn fill_from_wgpu(&mut self, wgpu_ctx: &WgpuCtx, texture: &wgpu::Texture) -> Result<(), WgpuError> {
        let size = texture.size();
        let res = Resolution { width: size.width as usize, height: size.height as usize };
        let (rgba, pitch) = read_back_padded(wgpu_ctx, texture);
        let s = self.ensure_size(res)?;
        self.ctx.check(unsafe { smr_surface_upload(self.ctx.raw, s, rgba.as_ptr().cast(), pitch) })?;
        self.filled = true;
        Ok(())
    }
}
impl Drop for HipNodeTexture {
    fn drop(&mut self) {
        self.clear()
    }
}

/// Texture -> mapped buffer with rows padded to 256 bytes, as OutputTexture / FramePreProcessor read back
/// (wgpu/texture/base.rs:84-118, state/frame_pre_processor.rs:157-181).  Returns the bytes and their row pitch.
fn read_back_padded(wgpu_ctx: &WgpuCtx, texture: &wgpu::Texture) -> (Vec<u8>, usize) {
    let buffer = texture.new_download_buffer(wgpu_ctx);
    let mut encoder = wgpu_ctx.device.create_command_encoder(&wgpu::CommandEncoderDescriptor { label: Some("HIP seam read-back") });
    texture.copy_to_buffer(&mut encoder, &buffer);
    wgpu_ctx.queue.submit(Some(encoder.finish()));
    let (s, r) = crossbeam_channel::bounded(1);
    buffer.slice(..).map_async(wgpu::MapMode::Read, move |result| {
        let _ = s.send(result);
    });
    while let Err(wgpu::PollError::Timeout) = wgpu_ctx.device.poll(wgpu::PollType::wait_indefinitely()) {}
    r.recv().unwrap().unwrap();
    let bytes = buffer.slice(..).get_mapped_range().unwrap().to_vec();
    buffer.unmap();
    let pitch = bytes.len() / (texture.size().height.max(1) as usize);
    (bytes, pitch)
}

/// What a layout or shader node samples: an input's raw frame (conversion fused into the resampler), a node surface, or nothing
/// (missing / stale input: the 1x1 transparent texture of layout.rs:204-212).
pub(crate) enum HipSource<'a> {
    Frame(&'a smr_frame),
    Surface(*const smr_surface, Resolution),
    None,
}

impl HipSource<'_> {
    pub(crate) fn resolution(&self) -> Option<Resolution> {
        match self {
            HipSource::Frame(f) => Some(Resolution { width: f.width as usize, height: f.height as usize }),
            HipSource::Surface(_, r) => Some(*r),
            HipSource::None => None,
        }
    }
    pub(crate) fn as_smr(&self) -> smr_source {
        match self {
            HipSource::Frame(f) => smr_source { kind: SMR_SOURCE_FRAME, surface: ptr::null(), frame: *f },
            HipSource::Surface(s, _) => smr_source { kind: SMR_SOURCE_SURFACE, surface: *s, frame: ptr::null() },
            HipSource::None => smr_source { kind: SMR_SOURCE_NONE, surface: ptr::null(), frame: ptr::null() },
        }
    }
}

/// Where a layout node renders to: the output's planar frame (root node: rgba_to_yuv is fused) or an RGBA8 node surface.
pub(crate) enum HipTarget<'a> {
    OutputFrame(&'a smr_frame),
    Node(*mut smr_surface),
}

/// OutputTexture::PlanarYuvTextures of an output (state/output_texture.rs:18-113) + its download.
pub(crate) struct HipOutput {
    ctx: Arc<HipCtx>,
    frame: smr_frame,
    format: OutputFrameFormat,
}
unsafe impl Send for HipOutput {}

impl HipOutput {
    pub(crate) fn new(ctx: Arc<HipCtx>, res: Resolution, format: OutputFrameFormat) -> Result<Self, WgpuError> {
        let f = match format {
            OutputFrameFormat::PlanarYuv420Bytes => SMR_FRAME_PLANAR_YUV420,
            OutputFrameFormat::PlanarYuv422Bytes => SMR_FRAME_PLANAR_YUV422,
            OutputFrameFormat::PlanarYuv444Bytes => SMR_FRAME_PLANAR_YUV444,
            OutputFrameFormat::RgbaWgpuTexture | OutputFrameFormat::Nv12WgpuTexture => {
                return Err(WgpuError::Validation("wgpu texture outputs are not available with the HIP rasteriser".into()));
            }
        };
        let mut frame = smr_frame { format: f, width: res.width as u32, height: res.height as u32, planes: [ptr::null_mut(); 3] };
        ctx.check(unsafe { smr_frame_create(ctx.raw, f, frame.width, frame.height, &mut frame) })?;
        Ok(Self { ctx, frame, format })
    }
    pub(crate) fn frame(&self) -> &smr_frame {
        &self.frame
    }
    pub(crate) fn resolution(&self) -> Resolution {
        Resolution { width: self.frame.width as usize, height: self.frame.height as usize }
    }
    /// start_download + download_buffer (output_texture.rs:68-113): tight planes, one blocking read-back.
    pub(crate) fn download(&self, pts: std::time::Duration) -> Result<Frame, WgpuError> {
        let (w, h) = (self.frame.width as usize, self.frame.height as usize);
        let (cw, ch) = match self.format {
            OutputFrameFormat::PlanarYuv422Bytes => (w / 2, h),
            OutputFrameFormat::PlanarYuv444Bytes => (w, h),
            _ => (w / 2, h / 2),
        };
        let (mut y, mut u, mut v) = (vec![0u8; w * h], vec![0u8; cw * ch], vec![0u8; cw * ch]);
        let host: [*mut std::os::raw::c_void; 3] = [y.as_mut_ptr().cast(), u.as_mut_ptr().cast(), v.as_mut_ptr().cast()];
        self.ctx.check(unsafe { smr_frame_download(self.ctx.raw, &self.frame, host.as_ptr()) })?;
        let planes = YuvPlanes { y_plane: Bytes::from(y), u_plane: Bytes::from(u), v_plane: Bytes::from(v) };
        let data = match self.format {
            OutputFrameFormat::PlanarYuv422Bytes => FrameData::PlanarYuv422(planes),
            OutputFrameFormat::PlanarYuv444Bytes => FrameData::PlanarYuv444(planes),
            _ => FrameData::PlanarYuv420(planes),
        };
        Ok(Frame { data, resolution: Resolution { width: w, height: h }, pts })
    }
    /// PlanarYuvTextures::fill_with_color(BLACK) (render_loop.rs:127-139): an output whose root is no layout node
    pub(crate) fn fill_black(&self) -> Result<(), WgpuError> {
        self.ctx.check(unsafe { smr_frame_fill_black(self.ctx.raw, &self.frame) }).map(|_| ())
    }
}
impl Drop for HipOutput {
    fn drop(&mut self) {
        unsafe { smr_frame_destroy(self.ctx.raw, &mut self.frame) }
    }
}

/// FramePreProcessor::process_to_bytes (state/frame_pre_processor.rs:84-107) on libsmr_hip: upload, smr_frame_preprocess.
pub(crate) fn preprocess_to_bytes(input: &mut HipInput, frame: Frame, resolution: Option<Resolution>) -> Result<Bytes, WgpuError> {
    let src = frame.resolution;
    input.upload(frame)?;
    let out = resolution.unwrap_or(src);
    let mut bytes = vec![0u8; out.width * out.height * 4];
    let (dw, dh) = match resolution {
        Some(r) => (r.width as u32, r.height as u32),
        None => (0, 0),
    };
    let ctx = input.ctx.clone();
    let f = input.frame().unwrap();
    ctx.check(unsafe { smr_frame_preprocess(ctx.raw, f, dw, dh, bytes.as_mut_ptr().cast(), 0) })?;
    Ok(Bytes::from(bytes))
}

/// The shaders smelter keeps in its own tree have hand-written counterparts in libsmr_hip (smr_builtin_shader); a registered WGSL
/// source is recognised by its text (FNV-1a 64 of the bytes, line endings normalised).  Anything else stays a wgpu-only shader.
pub(crate) fn builtin_shader_id(source: &str) -> Option<u32> {
    let mut h: u64 = 0xcbf2_9ce4_8422_2325;
    for b in source.bytes().filter(|b| *b != b'\r') {
        h ^= b as u64;
        h = h.wrapping_mul(0x0000_0100_0000_01b3);
    }
    match h {
@@SHADER_HASHES@@
        _ => None,
    }
}
"""

STATE_HIP_RS = r"""//! The render loop on libsmr_hip (SMELTER_HIP_DEVICE): populate_inputs / run_transforms / read_outputs of render_loop.rs with
//! the input conversion, every layout node and the output conversion on the HIP library.  See src/hip/mod.rs.
use std::{collections::HashMap, sync::Arc, time::Duration};

use crate::{
    Frame, FrameSet, InputId, OutputFrameFormat, OutputId, RenderingMode, Resolution,
    error::InitRendererEngineError,
    hip::{HipCtx, HipInput, HipNodeTexture, HipOutput, HipSource, HipTarget, sys::{smr_builtin_shader, smr_frame, smr_surface}},
    wgpu::WgpuError,
};

use super::{
    RenderCtx,
    node::{InnerRenderNode, RenderNode},
    render_graph::RenderGraph,
};

pub(super) struct HipState {
    ctx: Arc<HipCtx>,
    inputs: HashMap<InputId, HipInput>,
    outputs: HashMap<OutputId, HipOutputState>,
}

struct HipOutputState {
    output: HipOutput,
    /// node surfaces of the output's render tree, keyed by the node's position in it
    textures: HashMap<usize, HipNodeTexture>,
}

impl HipState {
    pub(super) fn new(device: i32, mode: RenderingMode, max_layouts: usize) -> Result<Self, InitRendererEngineError> {
        let ctx = HipCtx::new(device, mode, max_layouts).map_err(|err| InitRendererEngineError::HipInit(err.to_string()))?;
        Ok(Self { ctx, inputs: HashMap::new(), outputs: HashMap::new() })
    }

    /// RenderGraph::update (render_graph.rs:45-65): a new render tree for the output — its frame, no node surfaces yet.
    pub(super) fn update_output(&mut self, id: OutputId, resolution: Resolution, format: OutputFrameFormat) -> Result<(), WgpuError> {
        let output = HipOutput::new(self.ctx.clone(), resolution, format)?;
        self.outputs.insert(id, HipOutputState { output, textures: HashMap::new() });
        Ok(())
    }

    pub(super) fn render(&mut self, ctx: &mut RenderCtx, graph: &mut RenderGraph, mut frame_set: FrameSet<InputId>,
                         pts: Duration) -> Result<HashMap<OutputId, Frame>, WgpuError> {
        // populate_inputs (render_loop.rs:19-42): missing and stale frames leave the input without a texture
        self.inputs.retain(|id, _| graph.inputs.contains_key(id));
        for input_id in graph.inputs.keys() {
            let input = self.inputs.entry(input_id.clone()).or_insert_with(|| HipInput::new(self.ctx.clone()));
            let Some(frame) = frame_set.frames.remove(input_id) else {
                input.clear();
                continue;
            };
            // (stale: older than stream_fallback_timeout at this pts)
            let stale = frame.pts < frame_set.pts.saturating_sub(ctx.stream_fallback_timeout);
            if stale {
                input.clear();
            } else {
                input.upload(frame)?;
            }
        }
        // run_transforms + read_outputs (render_loop.rs:59-252), output by output
        self.outputs.retain(|id, _| graph.outputs.contains_key(id));
        let mut frames = HashMap::new();
        for (output_id, tree) in &mut graph.outputs {
            let Some(state) = self.outputs.get_mut(output_id) else { continue };
            let is_layout_root = matches!(tree.root.renderer, InnerRenderNode::Layout(_));
            let root_target = is_layout_root.then_some(state.output.frame());
            walk(&self.ctx, ctx, &self.inputs, &mut tree.root, 0, &mut state.textures, root_target, pts)?;
            if !is_layout_root {
                state.output.fill_black()?;
            }
            frames.insert(output_id.clone(), state.output.download(pts)?);
        }
        Ok(frames)
    }
}

fn source_of<'a>(node: &RenderNode, key: usize, inputs: &'a HashMap<InputId, HipInput>, textures: &'a HashMap<usize, HipNodeTexture>) -> HipSource<'a> {
    match &node.renderer {
        InnerRenderNode::InputStreamRef(id) => inputs.get(id).and_then(|i| i.frame()).map(HipSource::Frame).unwrap_or(HipSource::None),
        _ => textures.get(&key).and_then(|t| t.surface()).map(|(s, r)| HipSource::Surface(s as *const smr_surface, r)).unwrap_or(HipSource::None),
    }
}

/// Depth-first over the node tree (children before parents, as run_transforms does): the root layout node writes the output's
/// planar frame directly, every other node an RGBA8 surface.
#[allow(clippy::too_many_arguments)]
fn walk(hip: &Arc<HipCtx>, ctx: &mut RenderCtx, inputs: &HashMap<InputId, HipInput>, node: &mut RenderNode, key: usize,
        textures: &mut HashMap<usize, HipNodeTexture>, root_target: Option<&smr_frame>, pts: Duration) -> Result<(), WgpuError> {
    for (k, child) in node.children.iter_mut().enumerate() {
        walk(hip, ctx, inputs, child, key * 64 + k + 1, textures, None, pts)?;
    }
    let child_keys: Vec<usize> = (0..node.children.len()).map(|k| key * 64 + k + 1).collect();
    let RenderNode { output, renderer, children } = node;
    match renderer {
        InnerRenderNode::Layout(layout) => {
            let target = match root_target {
                Some(f) => HipTarget::OutputFrame(f),
                None => {
                    let res = layout.resolution_hip(pts);
                    HipTarget::Node(textures.entry(key).or_insert_with(|| HipNodeTexture::new(hip.clone())).ensure_size(res)?)
                }
            };
            let sources: Vec<HipSource> = children.iter().zip(&child_keys).map(|(c, k)| source_of(c, *k, inputs, textures)).collect();
            layout.render_hip(hip, &sources, target, pts)
        }
        InnerRenderNode::Text(text) => {
            // glyphon -> the node's wgpu texture (text_renderer.rs:72-167, once per node), read back once
            let tex = textures.entry(key).or_insert_with(|| HipNodeTexture::new(hip.clone()));
            if tex.filled {
                return Ok(());
            }
            text.render(ctx, output);
            match output.state() {
                Some(state) => tex.fill_from_wgpu(ctx.wgpu_ctx, state.texture()),
                None => Ok(()),
            }
        }
        InnerRenderNode::Image(image) => {
            image.render(ctx, output, pts);
            match output.state() {
                Some(state) => textures.entry(key).or_insert_with(|| HipNodeTexture::new(hip.clone())).fill_from_wgpu(ctx.wgpu_ctx, state.texture()),
                None => Ok(()),
            }
        }
        InnerRenderNode::Shader(shader) => {
            let Some(id) = shader.hip_builtin() else {
                return Err(WgpuError::Validation("this WGSL shader has no counterpart in libsmr_hip (smr_builtin_shader): it needs the wgpu renderer".into()));
            };
            let res = shader.resolution_hip();
            let dst = textures.entry(key).or_insert_with(|| HipNodeTexture::new(hip.clone())).ensure_size(res)?;
            let mut srcs: Vec<*const smr_surface> = Vec::with_capacity(children.len());
            for (c, k) in children.iter().zip(&child_keys) {
                match source_of(c, *k, inputs, textures) {
                    HipSource::Surface(s, _) => srcs.push(s),
                    HipSource::None => srcs.push(std::ptr::null()),
                    HipSource::Frame(_) => {
                        // (a raw input frame under a shader node: smelter always puts a layout node in between; keep the error explicit)
                        return Err(WgpuError::Validation("an input stream directly under a shader node is not supported by the HIP rasteriser".into()));
                    }
                }
            }
            let params = shader.hip_params();
            hip.check(unsafe {
                smr_builtin_shader(hip.raw(), id, params.as_ptr().cast(), params.len(), srcs.as_ptr(), srcs.len() as u32, dst, pts.as_secs_f32())
            }).map(|_| ())
        }
        InnerRenderNode::InputStreamRef(_) => Ok(()),
        InnerRenderNode::Web(_) => Err(WgpuError::Validation("web views need the wgpu renderer".into())),
    }
}
"""

LAYOUT_HIP_RS = r"""//! LayoutNode::render (transformations/layout.rs:169-232) on libsmr_hip: `layouts` + `flatten` stay the Rust above, the resampling
//! of scaled children and the layout shader are one smr_render_layouts call.  Lives here, under transformations::layout, because
//! RenderLayout, LayoutNode's fields and NestedLayout::flatten are private to this module.
use std::{ptr, time::Duration};

use crate::{
    RenderingMode, Resolution,
    hip::{HipCtx, HipSource, HipTarget, sys::*},
    scene::RGBAColor,
    wgpu::WgpuError,
};

use super::{LayoutNode, RenderLayout, RenderLayoutContent};

/// convert_to_shader_color (wgpu/utils.rs:51-81), f64 maths then cast — the same values ParamsBindGroups::update writes.
fn shader_color(mode: RenderingMode, c: &RGBAColor) -> [f32; 4] {
    let lin = |v: u8| {
        let v = v as f64 / 255.0;
        if v < 0.04045 { v / 12.92 } else { ((v + 0.055) / 1.055).powf(2.4) }
    };
    let a = c.3 as f64 / 255.0;
    let rgb = |v: u8| if mode == RenderingMode::CpuOptimized { v as f64 / 255.0 } else { lin(v) };
    [(a * rgb(c.0)) as f32, (a * rgb(c.1)) as f32, (a * rgb(c.2)) as f32, a as f32]
}

/// RenderLayout -> smr_layout, field for field in the order ParamsBindGroups::update packs them (layout/params.rs:223-303).
fn to_pod(mode: RenderingMode, l: &RenderLayout) -> smr_layout {
    let r = &l.border_radius;
    let mut pod = smr_layout {
        top: l.top,
        left: l.left,
        width: l.width,
        height: l.height,
        rotation_degrees: l.rotation_degrees,
        border_radius: [r.top_left, r.top_right, r.bottom_right, r.bottom_left],
        type_: 1,
        source_index: SMR_NO_SOURCE,
        color: [0.0; 4],
        border_color: [0.0; 4],
        border_width: 0.0,
        crop: [0.0; 4],
        blur_radius: 0.0,
        masks_len: l.masks.len().min(SMR_MAX_MASKS) as u32,
        masks: [smr_mask { radius: [0.0; 4], top: 0.0, left: 0.0, width: 0.0, height: 0.0 }; SMR_MAX_MASKS],
    };
    for (k, m) in l.masks.iter().take(SMR_MAX_MASKS).enumerate() {
        let mr = &m.radius;
        pod.masks[k] = smr_mask {
            radius: [mr.top_left, mr.top_right, mr.bottom_right, mr.bottom_left],
            top: m.top,
            left: m.left,
            width: m.width,
            height: m.height,
        };
    }
    match &l.content {
        RenderLayoutContent::Color { color, border_color, border_width } => {
            pod.type_ = 1;
            pod.color = shader_color(mode, color);
            pod.border_color = shader_color(mode, border_color);
            pod.border_width = *border_width;
        }
        RenderLayoutContent::ChildNode { index, border_color, border_width, crop } => {
            pod.type_ = 0;
            pod.source_index = *index as u32;
            pod.border_color = shader_color(mode, border_color);
            pod.border_width = *border_width;
            pod.crop = [crop.top, crop.left, crop.width, crop.height];
        }
        RenderLayoutContent::BoxShadow { color, blur_radius } => {
            pod.type_ = 2;
            pod.color = shader_color(mode, color);
            pod.blur_radius = *blur_radius;
        }
    }
    pod
}

impl LayoutNode {
    pub(crate) fn resolution_hip(&self, pts: Duration) -> Resolution {
        self.layout_provider.resolution(pts)
    }

    pub(crate) fn render_hip(&mut self, ctx: &HipCtx, sources: &[HipSource], target: HipTarget, pts: Duration) -> Result<(), WgpuError> {
        let resolutions: Vec<Option<Resolution>> = sources.iter().map(|s| s.resolution()).collect();
        let output_resolution = self.layout_provider.resolution(pts);
        let layouts = self.layout_provider.layouts(pts, &resolutions).flatten(&resolutions, output_resolution);
        // (more than max_layouts_count layouts: the reference logs and skips the rest, params.rs:176-182)
        let pod: Vec<smr_layout> = layouts.iter().take(ctx.max_layouts).map(|l| to_pod(ctx.mode, l)).collect();
        let srcs: Vec<smr_source> = sources.iter().map(HipSource::as_smr).collect();
        let (out_frame, out_rgba) = match target {
            HipTarget::OutputFrame(f) => (f as *const smr_frame, ptr::null_mut()),
            HipTarget::Node(s) => (ptr::null(), s),
        };
        ctx.check(unsafe {
            smr_render_layouts(ctx.raw(), pod.as_ptr(), pod.len() as u32, srcs.as_ptr(), srcs.len() as u32,
                               output_resolution.width as u32, output_resolution.height as u32, out_frame, out_rgba)
        }).map(|_| ())
    }
}
"""

# ---------------------------------------------------------------- anchored edits of reference files: (old, new) pairs
EDITS = {
    "src/lib.rs": [("pub mod error;\n", "pub mod error;\npub(crate) mod hip;\n")],
    "src/error.rs": [('''    #[error("Failed to initialize apply_layout transformation.")]
    LayoutTransformationsInitError(#[source] CreateShaderError),
}''', '''    #[error("Failed to initialize apply_layout transformation.")]
    LayoutTransformationsInitError(#[source] CreateShaderError),

    #[error("Failed to initialize the HIP rasteriser (SMELTER_HIP_DEVICE): {0}")]
    HipInit(String),
}''')],
    "src/state.rs": [
        ("mod render_loop;\n", "mod hip;\nmod render_loop;\n"),
        ('''    stream_fallback_timeout: Duration,

    wgpu_ctx: Arc<WgpuCtx>,
}''', '''    stream_fallback_timeout: Duration,

    wgpu_ctx: Arc<WgpuCtx>,

    /// Some(..) when SMELTER_HIP_DEVICE selects the HIP rasteriser (src/hip/mod.rs, src/state/hip.rs)
    hip: Option<hip::HipState>,
}'''),
        ('''        let wgpu_ctx = WgpuCtx::new(opts.device, opts.queue, opts.rendering_mode)?;

        Ok(Self {''', '''        let wgpu_ctx = WgpuCtx::new(opts.device, opts.queue, opts.rendering_mode)?;
        // SMELTER_HIP_DEVICE=<ordinal>: input conversion, layout nodes and output conversion run on libsmr_hip; the scene maths,
        // text and images are unchanged (wgpu keeps serving those two node kinds).
        let hip = match crate::hip::HipCtx::device_from_env() {
            Some(device) => Some(hip::HipState::new(device, opts.rendering_mode, opts.max_layouts_count)?),
            None => None,
        };

        Ok(Self {
            hip,'''),
        ('''        let pts = inputs.pts;
        trace!("Upload input textures");''', '''        let pts = inputs.pts;
        if let Some(hip) = &mut self.hip {
            // the same three steps on libsmr_hip (state/hip.rs); errors map onto the WgpuError classes (hip::HipCtx::check)
            let frames = hip.render(ctx, &mut self.render_graph, inputs, pts);
            scope.pop()?;
            return Ok(FrameSet { frames: frames?, pts });
        }
        trace!("Upload input textures");'''),
        ('''            output_node,
            output_format,
        )?;
        Ok(())''', '''            output_node,
            output_format,
        )?;
        if let Some(hip) = &mut self.hip {
            hip.update_output(output_id, resolution, output_format)?;
        }
        Ok(())'''),
    ],
    "src/state/frame_pre_processor.rs": [
        ('''    download_buffer: Option<(wgpu::Buffer, Resolution)>,
}''', '''    download_buffer: Option<(wgpu::Buffer, Resolution)>,
    /// SMELTER_HIP_DEVICE: `process_to_bytes` runs on libsmr_hip (smr_frame_preprocess); this is its device-side input frame.
    hip: Option<crate::hip::HipInput>,
}'''),
        ('''    pub fn new(wgpu_ctx: Arc<WgpuCtx>) -> Self {
        Self {
            wgpu_ctx,''', '''    pub fn new(wgpu_ctx: Arc<WgpuCtx>) -> Self {
        let hip = crate::hip::HipCtx::device_from_env()
            .and_then(|device| {
                crate::hip::HipCtx::new(device, wgpu_ctx.mode, crate::transformations::layout::DEFAULT_MAX_LAYOUTS_COUNT).ok()
            })
            .map(crate::hip::HipInput::new);
        Self {
            hip,
            wgpu_ctx,'''),
        ('''    ) -> bytes::Bytes {
        self.upload_and_convert_to_node_texture(frame);
''', '''    ) -> bytes::Bytes {
        if let Some(hip) = &mut self.hip {
            // (frames that are wgpu textures cannot be uploaded there: those fall through to the wgpu path)
            if let Ok(bytes) = crate::hip::preprocess_to_bytes(hip, frame.clone(), resolution) {
                return bytes;
            }
        }
        self.upload_and_convert_to_node_texture(frame);
'''),
    ],
    "src/transformations/layout.rs": [("mod flatten;\n", "mod flatten;\nmod hip;\n")],
    "src/transformations/shader.rs": [
        ('''pub struct Shader {
    pipeline: ShaderPipeline,
    clear_color: Option<wgpu::Color>,
}''', '''pub struct Shader {
    pipeline: ShaderPipeline,
    clear_color: Option<wgpu::Color>,
    /// Some(id): libsmr_hip has a hand-written counterpart of this WGSL source (hip::builtin_shader_id)
    pub(crate) hip_builtin: Option<u32>,
}'''),
        ('''        let clear_color = None;
        let pipeline = ShaderPipeline::new(wgpu_ctx, spec.source)?;

        Ok(Self {
            pipeline,
            clear_color,
        })''', '''        let clear_color = None;
        let hip_builtin = crate::hip::builtin_shader_id(&spec.source);
        let pipeline = ShaderPipeline::new(wgpu_ctx, spec.source)?;

        Ok(Self {
            pipeline,
            clear_color,
            hip_builtin,
        })'''),
    ],
    "src/transformations/shader/node.rs": [
        ('''    shader: Arc<Shader>,
    resolution: Resolution,
}''', '''    shader: Arc<Shader>,
    resolution: Resolution,
    /// the custom parameter bytes as the uniform buffer holds them (smr_builtin_shader takes the same bytes)
    hip_params: bytes::Bytes,
}'''),
        ('''            shader,
            resolution: *resolution,
        }''', '''            shader,
            resolution: *resolution,
            hip_params: shader_params.as_ref().map(|p| p.to_bytes()).unwrap_or_default(),
        }'''),
        ('''            self.shader.clear_color,
        )
    }
}''', '''            self.shader.clear_color,
        )
    }

    /// libsmr_hip's counterpart of this node's shader, if it is one of the in-tree WGSL sources (hip::builtin_shader_id)
    pub(crate) fn hip_builtin(&self) -> Option<u32> {
        self.shader.hip_builtin
    }

    pub(crate) fn hip_params(&self) -> &[u8] {
        &self.hip_params
    }

    pub(crate) fn resolution_hip(&self) -> Resolution {
        self.resolution
    }
}'''),
    ],
}

# WGSL sources the reference keeps in its tree -> smr_builtin_shader ids (include/smr.h)
BUILTIN_WGSL = [
    ("integration-tests/src/render_tests/yuv_tests/gradient.wgsl", "SMR_SHADER_GRADIENT"),
    ("integration-tests/src/render_tests/shader/red_border.wgsl", "SMR_SHADER_RED_BORDER"),
    ("integration-tests/src/render_tests/shader/circle_layout.wgsl", "SMR_SHADER_CIRCLE_LAYOUT"),
    ("integration-tests/src/render_tests/shader/fade_to_ball.wgsl", "SMR_SHADER_FADE_TO_BALL"),
    ("integration-tests/src/render_tests/shader/layout_planes.wgsl", "SMR_SHADER_LAYOUT_PLANES"),
    ("integration-tests/src/render_tests/shader/color_output_with_texture_count.wgsl", "SMR_SHADER_COLOR_BY_TEXTURE_COUNT"),
    ("integration-tests/examples/silly.wgsl", "SMR_SHADER_SILLY"),
    ("integration-tests/src/bin/benchmark/silly.wgsl", "SMR_SHADER_SILLY"),
]


def fnv1a64(data: bytes) -> int:
    h = 0xCBF29CE484222325
    for b in data:
        if b == 0x0D:
            continue
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def edit(text, pairs, path):
    for old, new in pairs:
        if text.count(old) != 1:
            raise SystemExit(f"{path}: the anchored text occurs {text.count(old)} times in the reference (expected once):\n{old[:200]}")
        text = text.replace(old, new, 1)
    return text


def udiff(a: str, b: str, rel: str) -> str:
    return "".join(difflib.unified_diff(a.splitlines(True), b.splitlines(True), f"a/{rel}", f"b/{rel}"))


def new_file(text: str, rel: str) -> str:
    lines = text.splitlines(True)
    return f"--- /dev/null\n+++ b/{rel}\n@@ -0,0 +1,{len(lines)} @@\n" + "".join("+" + ln for ln in lines)


BUILD_RS = '''fn main() {
    // libsmr_hip.so: built with `python -m smelter_amd.build` (hipcc, gfx950)
    if let Ok(dir) = std::env::var("SMR_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rerun-if-env-changed=SMR_HIP_LIB_DIR");
}
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    src = os.path.join(args.reference, "smelter-render")
    header = open(os.path.join(ROOT, "include", "smr.h")).read()
    parts = ["smelter-render on libsmr_hip (generated by tools/make_rust_patch.py of the smelter_amd repository; src/hip/sys.rs is generated\n"
             "from include/smr.h).  Apply inside the smelter checkout:  git apply smelter-render-hip.patch\n"
             "Build: SMR_HIP_LIB_DIR=<dir of libsmr_hip.so> cargo build -p smelter-render;  run with SMELTER_HIP_DEVICE=<ordinal>.\n\n"]

    def rd(rel):
        return open(os.path.join(src, rel)).read()

    for rel, pairs in EDITS.items():
        a = rd(rel)
        parts.append(udiff(a, edit(a, pairs, rel), "smelter-render/" + rel))
    seen, arms = set(), []
    for rel, const in BUILTIN_WGSL:
        h = fnv1a64(open(os.path.join(args.reference, rel), "rb").read())
        if h in seen:
            continue
        seen.add(h)
        arms.append(f"        0x{h:016x} => Some({const}), // {rel}")
    mod_rs = MOD_RS.lstrip("\n").replace("@@SHADER_HASHES@@", "\n".join(arms))
    parts.append(new_file(mod_rs, "smelter-render/src/hip/mod.rs"))
    parts.append(new_file(gen_sys_rs(header), "smelter-render/src/hip/sys.rs"))
    parts.append(new_file(STATE_HIP_RS.lstrip("\n"), "smelter-render/src/state/hip.rs"))
    parts.append(new_file(LAYOUT_HIP_RS.lstrip("\n"), "smelter-render/src/transformations/layout/hip.rs"))
    parts.append(new_file(BUILD_RS, "smelter-render/build.rs"))
    out = os.path.join(ROOT, "integration", "smelter-render-hip.patch")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    open(out, "w").write("".join(parts))
    print(out, len("".join(parts).splitlines()), "lines;", len(c_decls(header)), "entry points bound")


if __name__ == "__main__":
    main()
