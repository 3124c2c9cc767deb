"""integration/smelter-render-hip.patch — the Rust seam a maintainer applies to smelter-render.

No cargo in this environment: the patch is reviewed text.  Two groups of tests keep it honest:
* against include/smr.h (always): the `unsafe extern "C"` block binds every SMR_API entry point under the same name, with the same
  arity and pointer shapes; the #[repr(C)] mirrors have the header's fields in the header's order; every enum constant is generated.
* against the reference checkout (when /root/reference is present — it is where the CPU suite runs, not on the GPU box): the patch
  applies, and with it applied every crate item the ADDED Rust names exists and is visible from the module that names it
  (tools/rust_index.py: modules, visibility, fields, variants, methods, re-exports): `use` paths, `Type::item` paths, bare type
  names, free-function calls, `.member` accesses, the members the seam relies on with their parameter counts, the matches that have
  no wildcard arm, the constructors of structs that gained a field.  tools/make_rust_patch.py regenerates the patch."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tools"))

needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "smelter-render", "src")), reason="the reference checkout is not present")


def _header_text():
    return re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", "smr.h")).read(), flags=re.S)


def _header_decls():
    out = {}
    for m in re.finditer(r"SMR_API\s+([^;(]+?)\b(smr_\w+)\s*\(([^;]*?)\)\s*;", _header_text(), flags=re.S):
        params = " ".join(m.group(3).split())
        plist = [] if params in ("void", "") else [p.strip() for p in params.split(",")]
        out[m.group(2)] = (m.group(1).strip(), plist)
    return out


def _patch_text():
    return open(os.path.join(ROOT, "integration", "smelter-render-hip.patch")).read()


def _added_by_file():
    """{path relative to the checkout: [added lines]} of the patch."""
    out, cur = {}, None
    for ln in _patch_text().splitlines():
        if ln.startswith("+++ b/"):
            cur = out.setdefault(ln[6:], [])
        elif ln.startswith("+") and not ln.startswith("+++") and cur is not None:
            cur.append(ln[1:])
    return out


def _added():
    return "\n".join("\n".join(v) for v in _added_by_file().values())


def _rust_decls():
    added = _added()
    block = added[added.index('unsafe extern "C" {'):]
    block = block[: block.index("\n}")]
    out = {}
    for m in re.finditer(r"pub fn (smr_\w+)\((.*?)\)( -> ([^;]+))?;", block):
        params = [p.strip() for p in m.group(2).split(",")] if m.group(2).strip() else []
        out[m.group(1)] = (m.group(4) or "()", params)
    return out


# --------------------------------------------------------------------------------------------------------- against include/smr.h
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


def test_rust_struct_mirrors_have_the_c_fields_in_order():
    hdr = _header_text()
    added = _added()
    for name in ("smr_frame", "smr_mask", "smr_layout", "smr_source", "smr_glyph", "smr_resample_plan", "smr_scene_node", "smr_input_frame",
                 "smr_output_frame", "smr_surface_info", "smr_gaussian_blur_params", "smr_circle_layout", "smr_text_params", "smr_text_run"):
        c_body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", hdr, flags=re.S).group(1)
        c_fields = []
        for decl in c_body.split(";"):
            if not decl.strip():
                continue
            names = [re.search(r"(\w+)\s*(\[[^\]]*\])*\s*$", d.strip()).group(1) for d in decl.split(",") if d.strip()]
            c_fields += names
        r_body = re.search(r"pub struct " + name + r" \{(.*?)\n?\}", added, flags=re.S).group(1)
        r_body = "\n".join(ln.split("//")[0] for ln in r_body.split("\n"))
        r_fields = [f.rstrip("_") for f in re.findall(r"pub (\w+):", r_body)]
        assert c_fields == r_fields, (name, c_fields, r_fields)
    # every struct the binding's signatures mention is declared (opaque or mirrored)
    declared = set(re.findall(r"pub (?:struct|type) (\w+)", added))
    for _ret, params in _rust_decls().values():
        for p in params:
            for t in re.findall(r"\bsmr_\w+", p.split(": ", 1)[1]):
                assert t in declared, (t, p)


def test_enum_constants_are_the_header_values():
    added = _added()
    n = 0
    for m in re.finditer(r"typedef\s+enum\s+(\w+)\s*\{(.*?)\}", _header_text(), flags=re.S):
        nxt = 0
        for item in m.group(2).split(","):
            if not item.strip():
                continue
            name, _, val = item.strip().partition("=")
            nxt = int(val.strip(), 0) if val.strip() else nxt
            assert re.search(r"pub const %s: (c_int|u32) = %d;" % (name.strip(), nxt), added), name
            nxt += 1
            n += 1
    assert n >= 40


def test_patch_hygiene():
    p, added = _patch_text(), _added()
    files = _added_by_file()
    for rel in ("smelter-render/src/state.rs", "smelter-render/src/lib.rs", "smelter-render/src/error.rs", "smelter-render/src/state/frame_pre_processor.rs",
                "smelter-render/src/transformations/layout.rs", "smelter-render/src/transformations/shader.rs",
                "smelter-render/src/transformations/shader/node.rs", "smelter-render/src/hip/mod.rs", "smelter-render/src/hip/sys.rs",
                "smelter-render/src/state/hip.rs", "smelter-render/src/transformations/layout/hip.rs", "smelter-render/build.rs"):
        assert rel in files, rel
    # no enum variant, no new field on a struct other crates construct: RenderingMode is matched exhaustively all over the workspace and
    # RendererOptions is built by smelter-core, the integration tests and the wasm crate
    assert "smelter-render/src/types.rs" not in files
    assert "RenderingMode::Hip" not in added and "hip_device:" not in "\n".join(files["smelter-render/src/state.rs"])
    # edition 2024 (smelter-render/Cargo.toml): extern blocks are `unsafe extern`
    assert 'unsafe extern "C" {' in added and not re.search(r'^\s*extern "C" \{', added, flags=re.M)
    # wgpu's mapped range is a Result in the version the workspace pins (state/frame_pre_processor.rs:195)
    assert "get_mapped_range().unwrap()" in added and not re.search(r"get_mapped_range\(\)\.(?!unwrap)", added)
    for needle in ("SMELTER_HIP_DEVICE", "fn render_hip", "smr_render_layouts(ctx.raw()", "smr_frame_download(self.ctx.raw", "smr_frame_preprocess(ctx.raw",
                   "smr_builtin_shader(hip.raw()", "WgpuError::Validation", "HipInit(String)"):
        assert needle in added, needle
    assert p.count("\n--- /dev/null") == 5


# ------------------------------------------------------------------------------------------------- against the reference checkout
@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    from rust_index import Crate

    d = tmp_path_factory.mktemp("smelter")
    shutil.copytree(os.path.join(REF, "smelter-render"), os.path.join(d, "smelter-render"), ignore=shutil.ignore_patterns("target", "*.png", "*.jpg"))
    r = subprocess.run(["git", "apply", os.path.join(ROOT, "integration", "smelter-render-hip.patch")], cwd=d, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    crate = Crate(os.path.join(d, "smelter-render", "src"))
    by_file = {os.path.relpath(m.file, str(d)): m for m in crate.modules.values()}
    return crate, by_file


def _code_only(text):
    """Added Rust with comments, string and char literals blanked."""
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r'"(?:[^"\\]|\\.)*"', '""', text)
    text = re.sub(r"b?'(?:[^'\\]|\\.)'", "' '", text)
    text = re.sub(r"#!?\[[^\]]*\]", "", text)   # attributes
    return text


def _added_modules(patched):
    crate, by_file = patched
    for rel, lines in _added_by_file().items():
        if rel.endswith(".rs") and rel in by_file:
            yield by_file[rel], _code_only("\n".join(lines))


STD_TYPES = {"Some", "None", "Ok", "Err", "Vec", "Option", "Result", "String", "Box", "Self", "Send", "Sync", "Drop", "From", "Default", "Clone",
             "Copy", "Debug", "Into", "Iterator", "FnOnce", "Fn", "FnMut"}
EXTERNAL_CRATES = {"std", "core", "bytes", "wgpu", "crossbeam_channel", "tracing"}   # smelter-render/Cargo.toml [dependencies] + std
KEYWORDS = {"if", "match", "while", "for", "return", "loop", "fn", "let", "in", "as", "move", "unsafe", "else", "impl", "use", "pub", "mut", "ref",
            "where", "struct", "enum", "mod", "crate", "super", "self", "continue", "break", "const", "static", "type", "dyn", "true", "false"}
# members of std / wgpu / bytes types the added code calls (reviewed by hand: none of these is a smelter-render item)
STD_MEMBERS = {
    "as_ptr", "as_mut_ptr", "cast", "len", "iter", "iter_mut", "map", "collect", "clone", "unwrap", "ok", "and_then", "to_string", "into", "is_some",
    "as_ref", "as_mut", "take", "min", "max", "enumerate", "zip", "filter", "bytes", "wrapping_mul", "to_vec", "unmap", "slice", "map_async",
    "get_mapped_range", "send", "recv", "poll", "submit", "finish", "create_command_encoder", "size", "width", "height", "insert", "get", "get_mut",
    "entry", "or_insert_with", "retain", "contains_key", "keys", "remove", "parse", "to_string_lossy", "into_owned", "map_err", "unwrap_or",
    "unwrap_or_default", "then_some", "push", "as_secs_f32", "saturating_sub", "powf", "is_null", "lstrip", "to_le_bytes", "unwrap_or_else",
    "device", "queue",
}


def _scope_has(crate, mod, name):
    hit = crate._lookup_in_scope(name, mod.path, mod.path, 0)   # None: unknown; (None, why): exists but not visible
    return hit is not None and hit[0] is not None


@needs_reference
def test_use_paths_of_the_added_code_resolve_and_are_visible(patched):
    from rust_index import expand_use

    crate, _ = patched
    checked = 0
    for mod, code in _added_modules(patched):
        for m in re.finditer(r"(?:^|\n)\s*(?:pub(?:\(\w+\))?\s+)?use\s+(.*?);", code, flags=re.S):
            for path, _alias in expand_use(m.group(1)):
                if path[-1] == "*":
                    path = path[:-1]
                target, why = crate.resolve(path, mod.path)
                if target in ("external", "imported-external"):
                    assert target == "imported-external" or path[0] in EXTERNAL_CRATES, (mod.file, path)
                    continue
                assert target is not None, (mod.file, "::".join(path), why)
                checked += 1
    assert checked >= 30


@needs_reference
def test_paths_and_type_names_of_the_added_code_resolve(patched):
    crate, _ = patched
    checked = 0
    for mod, code in _added_modules(patched):
        if mod.path[-1] == "sys":
            continue  # the generated binding: checked against the header above
        code = re.sub(r"(?:^|\n)\s*(?:pub(?:\(\w+\))?\s+)?use\s+.*?;", "\n", code, flags=re.S)
        generics = set(re.findall(r"<'?(\w+)>", code))
        declared_here = {v for it in mod.items.values() for v in it.variants}   # (a variant's own declaration line)
        for m in re.finditer(r"(?<![\w:])((?:\w+::)+\w+)", code):
            path = tuple(m.group(1).split("::"))
            if path[0] in ("u8", "u32", "u64", "i32", "f32", "f64", "usize", "Self"):
                continue
            target, why = crate.resolve(path, mod.path)
            if target == "external":
                assert path[0] in EXTERNAL_CRATES | STD_TYPES, (mod.file, m.group(1))
                continue
            assert target is not None, (mod.file, m.group(1), why)
            checked += 1
        for m in re.finditer(r"(?<![\w:.])([A-Z]\w*)\b(?!::)", code):
            name = m.group(1)
            if name in STD_TYPES or name in generics or name in declared_here:
                continue
            assert _scope_has(crate, mod, name), (mod.file, name, "is not in scope")
            checked += 1
    assert checked >= 150


@needs_reference
def test_free_functions_called_by_the_added_code_are_in_scope(patched):
    crate, _ = patched
    checked = 0
    for mod, code in _added_modules(patched):
        if mod.path[-1] == "sys":
            continue
        local_closures = set(re.findall(r"let (\w+) = (?:move )?\|", code))
        for m in re.finditer(r"(?<![\w:.!])([a-z_]\w*)\(", code):
            name = m.group(1)
            before = code[max(0, m.start() - 4): m.start()]
            if name in KEYWORDS or before.endswith("fn ") or name in local_closures or name in ("pub", "crate", "super"):
                continue
            assert _scope_has(crate, mod, name), (mod.file, name, "is not a function in scope")
            checked += 1
    assert checked >= 25


@needs_reference
def test_members_touched_by_the_added_code_exist_and_are_visible(patched):
    crate, _ = patched
    checked = 0
    for mod, code in _added_modules(patched):
        if mod.path[-1] == "sys":
            continue
        allowed = crate.all_member_names(mod.path) | STD_MEMBERS
        for m in re.finditer(r"\.([a-z_]\w*)\b", code):
            name = m.group(1)
            if code[m.start() - 1].isdigit() and code[m.start() + 1:m.start() + 2].isdigit():
                continue
            assert name in allowed, (mod.file, name, "no such field or method is visible from this module")
            checked += 1
    assert checked >= 200


def _member(crate, type_name, member, from_mod, want_method=False):
    """(kind, n_params or None) of type_name's field / method / variant `member` if it is visible from from_mod."""
    items = crate.find_type(type_name)
    assert items, type_name
    for it in items:
        if member in it.fields and not want_method:
            assert crate.visible(it.fields[member], it.module, from_mod), (type_name, member, it.fields[member], from_mod)
            return "field", None
        if member in it.variants:
            return "variant", None
        meth = crate.methods_of(it).get(member)
        if meth is not None:
            assert crate.visible(meth[0], meth[3], from_mod), (type_name, member, meth, from_mod)
            return "method", meth[1]
    raise AssertionError(f"{type_name} has no member `{member}`")


STATE_HIP = ("crate", "state", "hip")
LAYOUT_HIP = ("crate", "transformations", "layout", "hip")
HIP = ("crate", "hip")

# what the seam relies on, with the parameter counts its calls pass (None: a field or variant)
RELIED_ON = [
    (STATE_HIP, "RenderNode", "output", None), (STATE_HIP, "RenderNode", "renderer", None), (STATE_HIP, "RenderNode", "children", None),
    (STATE_HIP, "RenderGraph", "inputs", None), (STATE_HIP, "RenderGraph", "outputs", None), (STATE_HIP, "OutputRenderTree", "root", None),
    (STATE_HIP, "RenderCtx", "wgpu_ctx", None), (STATE_HIP, "RenderCtx", "stream_fallback_timeout", None),
    (STATE_HIP, "FrameSet", "frames", None), (STATE_HIP, "FrameSet", "pts", None), (STATE_HIP, "Frame", "pts", None),
    (STATE_HIP, "TextRendererNode", "render", 2), (STATE_HIP, "ImageNode", "render", 3), (STATE_HIP, "NodeTexture", "state", 0),
    (STATE_HIP, "NodeTextureState", "texture", 0), (STATE_HIP, "ShaderNode", "hip_builtin", 0), (STATE_HIP, "ShaderNode", "hip_params", 0),
    (STATE_HIP, "ShaderNode", "resolution_hip", 0), (STATE_HIP, "LayoutNode", "resolution_hip", 1), (STATE_HIP, "LayoutNode", "render_hip", 4),
    (STATE_HIP, "HipInput", "upload", 1), (STATE_HIP, "HipInput", "clear", 0), (STATE_HIP, "HipInput", "frame", 0),
    (STATE_HIP, "HipNodeTexture", "ensure_size", 1), (STATE_HIP, "HipNodeTexture", "fill_from_wgpu", 2), (STATE_HIP, "HipNodeTexture", "filled", None),
    (STATE_HIP, "HipOutput", "download", 1), (STATE_HIP, "HipOutput", "fill_black", 0), (STATE_HIP, "HipCtx", "check", 1), (STATE_HIP, "HipCtx", "raw", 0),
    (STATE_HIP, "InitRendererEngineError", "HipInit", None),
    (("crate", "state"), "HipState", "render", 4), (("crate", "state"), "HipState", "update_output", 3), (("crate", "state"), "HipState", "new", 3),
    (("crate", "state"), "RendererOptions", "rendering_mode", None), (("crate", "state"), "RendererOptions", "max_layouts_count", None),
    (LAYOUT_HIP, "LayoutNode", "layout_provider", None), (LAYOUT_HIP, "LayoutProvider", "layouts", 2), (LAYOUT_HIP, "LayoutProvider", "resolution", 1),
    (LAYOUT_HIP, "NestedLayout", "flatten", 2), (LAYOUT_HIP, "HipCtx", "max_layouts", None), (LAYOUT_HIP, "HipCtx", "mode", None),
    (LAYOUT_HIP, "HipSource", "resolution", 0), (LAYOUT_HIP, "HipSource", "as_smr", 0),
    *[(LAYOUT_HIP, "RenderLayout", f, None) for f in ("top", "left", "width", "height", "rotation_degrees", "border_radius", "masks", "content")],
    *[(LAYOUT_HIP, "Mask", f, None) for f in ("radius", "top", "left", "width", "height")],
    *[(LAYOUT_HIP, "Crop", f, None) for f in ("top", "left", "width", "height")],
    *[(LAYOUT_HIP, "BorderRadius", f, None) for f in ("top_left", "top_right", "bottom_right", "bottom_left")],
    (HIP, "WgpuCtx", "mode", None), (HIP, "WgpuCtx", "device", None), (HIP, "WgpuCtx", "queue", None),
    (HIP, "TextureExt", "new_download_buffer", 1), (HIP, "TextureExt", "copy_to_buffer", 2),
    *[(HIP, "WgpuError", v, None) for v in ("Validation", "OutOfMemory", "Internal")],
    *[(HIP, "YuvPlanes", f, None) for f in ("y_plane", "u_plane", "v_plane")], (HIP, "NvPlanes", "y_plane", None), (HIP, "NvPlanes", "uv_planes", None),
    (HIP, "Frame", "data", None), (HIP, "Frame", "resolution", None), (HIP, "Resolution", "width", None), (HIP, "Resolution", "height", None),
    (("crate", "state", "frame_pre_processor"), "HipCtx", "new", 3), (("crate", "state", "frame_pre_processor"), "HipCtx", "device_from_env", 0),
    (("crate", "state", "frame_pre_processor"), "HipInput", "new", 1),
    (("crate", "transformations", "shader", "node"), "Shader", "hip_builtin", None),
    (("crate", "transformations", "shader", "node"), "ShaderParamExt", "to_bytes", 0),
]


@needs_reference
def test_members_the_seam_relies_on(patched):
    crate, _ = patched
    for from_mod, type_name, member, n in RELIED_ON:
        kind, params = _member(crate, type_name, member, from_mod, want_method=n is not None)
        if n is not None:
            assert kind == "method" and params == n, (type_name, member, params, n)
    # free functions of the adapter with the argument counts their callers pass
    hip = crate.modules[HIP]
    assert hip.items["preprocess_to_bytes"].methods["()"][1] == 3 and hip.items["builtin_shader_id"].methods["()"][1] == 1
    # tuple struct RGBAColor(u8, u8, u8, u8): c.0 .. c.3
    assert crate.find_type("RGBAColor")[0].tuple_struct
    # struct-variant fields the layout conversion destructures (transformations/layout.rs)
    text = crate.modules[("crate", "transformations", "layout")].text
    body = re.search(r"enum RenderLayoutContent \{(.*?)\n\}", text, flags=re.S).group(1)
    for variant, fields in (("Color", ("color", "border_color", "border_width")), ("ChildNode", ("index", "border_color", "border_width", "crop")),
                            ("BoxShadow", ("color", "blur_radius"))):
        vb = re.search(variant + r" \{(.*?)\}", body, flags=re.S).group(1)
        assert tuple(re.findall(r"(\w+):", vb)) == fields, (variant, vb)


@needs_reference
def test_matches_without_a_wildcard_name_every_variant(patched):
    crate, _ = patched
    added = {m.path: code for m, code in _added_modules(patched)}
    for mod_path, enum, prefix in ((HIP, "FrameData", "FrameData::"), (HIP, "RenderingMode", "RenderingMode::"), (HIP, "OutputFrameFormat", "OutputFrameFormat::"),
                                   (STATE_HIP, "InnerRenderNode", "InnerRenderNode::"), (LAYOUT_HIP, "RenderLayoutContent", "RenderLayoutContent::")):
        variants = crate.resolve((enum,), mod_path)[0].variants   # the enum this module's scope gives the name to
        assert len(variants) >= 3
        named = set(re.findall(re.escape(prefix) + r"(\w+)", added[mod_path]))
        assert named >= set(variants), (enum, set(variants) - named)
        assert named <= set(variants), (enum, named - set(variants))


@needs_reference
def test_destructuring_and_constructors_cover_the_fields(patched):
    crate, by_file = patched
    node = crate.find_type("RenderNode")[0]
    pat = re.search(r"let RenderNode \{(.*?)\} = node;", _added()).group(1)
    assert {f.strip() for f in pat.split(",")} == set(node.fields)
    # the structs that gained a field have one constructor each, and it sets the field
    for rel, struct, new_field in (("smelter-render/src/state.rs", "InnerRenderer", "hip"), ("smelter-render/src/state/frame_pre_processor.rs", "FramePreProcessor", "hip"),
                                   ("smelter-render/src/transformations/shader.rs", "Shader", "hip_builtin"),
                                   ("smelter-render/src/transformations/shader/node.rs", "ShaderNode", "hip_params")):
        mod = by_file[rel]
        assert new_field in crate.find_type(struct)[0].fields, (struct, new_field)
        literals = re.findall(r"\bSelf \{(.*?)\n\s*\}\)?", mod.text, flags=re.S)
        fields = set(crate.find_type(struct)[0].fields)
        ctor = [lit for lit in literals if len(fields & set(re.findall(r"(\w+)[,:\n]", lit))) >= len(fields) - 1]
        assert len(ctor) == 1 and re.search(r"\b%s\b" % new_field, ctor[0]), (struct, len(ctor))
        # (its private fields allow a struct literal only in the defining module and below: no other literal there)
        home = crate.find_type(struct)[0].module
        for m2 in crate.modules.values():
            if m2.path[: len(home)] == home:
                named = [ln for ln in m2.text.split("\n") if re.search(r"(?<![:\w])%s \{" % struct, ln) and not re.search(r"\b(struct|impl|for) ", ln)]
                assert not named, (struct, m2.file, named)
    # smr_layout { .. } in to_pod names every field of the mirror
    mirror = re.search(r"pub struct smr_layout \{(.*?)\n\}", _added(), flags=re.S).group(1)
    mirror = "\n".join(ln.split("//")[0] for ln in mirror.split("\n"))
    lit = re.search(r"let mut pod = smr_layout \{(.*?)\n    \};", _added(), flags=re.S).group(1)
    assert set(re.findall(r"pub (\w+):", mirror)) == set(re.findall(r"^\s{8}(\w+):", lit, flags=re.M))


@needs_reference
def test_builtin_shader_hashes_are_the_sources_in_the_checkout():
    """hip::builtin_shader_id recognises the reference's own WGSL files by FNV-1a 64 of their text."""
    from make_rust_patch import BUILTIN_WGSL, fnv1a64

    added = _added()
    for rel, const in BUILTIN_WGSL:
        h = fnv1a64(open(os.path.join(REF, rel), "rb").read())
        assert f"0x{h:016x} => Some({const})" in added, rel
