"""integration/smelter-render-hip.patch against include/smr.h: the Rust `extern "C"` block of the patch binds
every SMR_API entry point, under the same name, with the same number of parameters and matching pointer shapes, and the
patch touches the seam files the reference dispatches through.  (No cargo here: the patch is reviewed text; this keeps its
binding from drifting off the header.  tools/make_rust_patch.py regenerates it.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    text = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "smr.h")).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"SMR_API\s+([^;(]+?)\b(smr_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        params = " ".join(m.group(3).split())
        plist = [] if params in ("void", "") else [p.strip() for p in params.split(",")]
        out[m.group(2)] = (m.group(1).strip(), plist)
    return out


def _patch_text():
    return open(os.path.join(ROOT, "integration", "smelter-render-hip.patch")).read()


def _rust_decls():
    added = "\n".join(ln[1:] for ln in _patch_text().splitlines() if ln.startswith("+") and not ln.startswith("+++"))
    block = added[added.index('extern "C" {'):]
    block = block[: block.index("\n}")]
    out = {}
    for m in re.finditer(r"pub fn (smr_\w+)\((.*?)\)( -> ([^;]+))?;", block):
        params = [p.strip() for p in m.group(2).split(",")] if m.group(2).strip() else []
        out[m.group(1)] = (m.group(4) or "()", params)
    return out


def test_every_header_symbol_is_bound_with_the_same_arity():
    c, r = _header_decls(), _rust_decls()
    assert len(c) >= 70
    assert set(c) == set(r), set(c) ^ set(r)
    for name, (ret, params) in c.items():
        rret, rparams = r[name]
        assert len(params) == len(rparams), name
        # pointer depth per parameter (a C array parameter is a pointer)
        for cp, rp in zip(params, rparams):
            depth_c = cp.count("*") + (1 if "[" in cp else 0)
            depth_r = rp.count("*const") + rp.count("*mut")
            assert depth_c == depth_r, (name, cp, rp)
            if depth_c:
                innermost_const_c = re.match(r"\s*const\b", cp) is not None
                innermost_const_r = re.search(r"\*(const|mut) [a-z_0-9]+$", rp.split(": ", 1)[1]).group(1) == "const"
                assert innermost_const_c == innermost_const_r, (name, cp, rp)
        assert (ret == "void") == (rret == "()"), name
        assert ret.count("*") == rret.count("*"), name


def test_patch_routes_the_reference_seam():
    p = _patch_text()
    for rel in ("smelter-render/src/types.rs", "smelter-render/src/state.rs", "smelter-render/src/state/node.rs", "smelter-render/src/lib.rs",
                "smelter-render/src/hip/mod.rs", "smelter-render/src/hip/sys.rs", "smelter-render/build.rs"):
        assert f"+++ b/{rel}" in p, rel
    for needle in ("Hip,", "populate_inputs_hip", "render_output_hip", "fn render_hip", "smr_render_layouts(ctx.raw", "smr_frame_download(ctx.raw",
                   "WgpuError::Validation", "RenderingMode::Hip =>"):
        assert needle in p, needle


def test_rust_struct_mirrors_match_the_c_layout():
    """Field counts of the #[repr(C)] mirrors against the header's struct definitions."""
    hdr = open(os.path.join(ROOT, "include", "smr.h")).read()
    added = "\n".join(ln[1:] for ln in _patch_text().splitlines() if ln.startswith("+") and not ln.startswith("+++"))
    for name in ("smr_frame", "smr_mask", "smr_layout", "smr_source", "smr_glyph", "smr_resample_plan", "smr_scene_node", "smr_input_frame",
                 "smr_output_frame", "smr_surface_info"):
        c_body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", hdr, flags=re.S).group(1)
        c_body = re.sub(r"/\*.*?\*/", " ", c_body, flags=re.S)
        c_fields = sum(len([d for d in decl.split(",") if d.strip()]) for decl in c_body.split(";") if decl.strip())
        r_body = re.search(r"pub struct " + name + r" \{(.*?)\n?\}", added, flags=re.S).group(1)
        r_fields = len(re.findall(r"pub \w+:", r_body))
        assert c_fields == r_fields, (name, c_fields, r_fields)
