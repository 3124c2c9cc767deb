#!/usr/bin/env python3
"""A small item index of a rustfmt-formatted Rust crate: modules, items with their visibility, struct fields, enum variants, inherent
and trait methods (with parameter counts), `use` declarations (re-exports included) — enough to check, without rustc, that code added
to the crate names only things that exist and that it is allowed to see.  Used by tests/test_rust_patch.py on the reference's
smelter-render with integration/smelter-render-hip.patch applied (there is no cargo in this environment).

Not a Rust parser: it relies on rustfmt's layout (module-level items start at column 0, members at one indentation step) and on the
subset of the language the crate and the patch use.  What it cannot decide it reports as unknown rather than guessing.
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass, field

VIS = r"(?:pub(?:\((?:crate|super|self|in [\w:]+)\))?\s+)?"
ITEM_RE = re.compile(r"^(" + VIS + r")(?:async\s+|const\s+|unsafe\s+|extern\s+\"C\"\s+)*(struct|enum|trait|fn|const|static|type|mod|union)\s+(\w+)")
MEMBER_FN_RE = re.compile(r"^\s+(" + VIS + r")(?:async\s+|const\s+|unsafe\s+)*fn\s+(\w+)")
FIELD_RE = re.compile(r"^\s+(" + VIS + r")(\w+):")
VARIANT_RE = re.compile(r"^\s+(\w+)\s*(?:[({,=]|$)")
IMPL_RE = re.compile(r"^(?:unsafe\s+)?impl(?:<[^>]*>)?\s+(?:([\w:<>', ]+?)\s+for\s+)?([\w:]+)")


@dataclass
class Item:
    kind: str
    name: str
    vis: str            # "", "pub", "pub(crate)", "pub(super)", "pub(in a::b)"
    module: tuple       # module path the item is declared in, e.g. ("crate", "state", "node")
    fields: dict = field(default_factory=dict)     # struct: name -> vis
    variants: list = field(default_factory=list)   # enum
    methods: dict = field(default_factory=dict)    # name -> (vis, n_params without self, has_self)
    tuple_struct: bool = False


@dataclass
class Module:
    path: tuple
    file: str
    items: dict = field(default_factory=dict)      # name -> Item
    uses: list = field(default_factory=list)       # (vis, path tuple, alias or None)  with "*" as the last element for globs
    text: str = ""


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    out = []
    for ln in text.split("\n"):
        # (string literals containing // are rare in declarations; good enough for the index)
        k = ln.find("//")
        out.append(ln if k < 0 else ln[:k])
    return "\n".join(out)


def expand_use(tree: str):
    """`a::{b, c::{d, e as f}, g::*}` -> [(("a","b"),None), (("a","c","d"),None), (("a","c","e"),"f"), (("a","g","*"),None)]"""
    tree = " ".join(tree.split())

    def split_top(s):
        parts, depth, cur = [], 0, ""
        for ch in s:
            if ch == "{":
                depth += 1
            elif ch == "}":
                depth -= 1
            if ch == "," and depth == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append(cur)
        return [p.strip() for p in parts if p.strip()]

    def rec(prefix, s):
        s = s.strip()
        k = s.find("{")
        if k < 0:
            alias = None
            if " as " in s:
                s, alias = [t.strip() for t in s.split(" as ")]
            segs = tuple(t for t in s.split("::") if t)
            if segs and segs[-1] == "self":
                segs = segs[:-1]
            return [(prefix + segs, alias)]
        head = tuple(t for t in s[:k].split("::") if t)
        inner = s[k + 1: s.rindex("}")]
        out = []
        for part in split_top(inner):
            out += rec(prefix + head, part)
        return out

    return rec((), tree)


def fn_paren(text: str) -> int:
    """Index of the parameter list's `(` of the first fn in text (not the one of `pub(crate)`)."""
    m = re.search(r"\bfn\s+\w+\s*(<[^(]*>)?\s*\(", text)
    return m.end() - 1


def count_params(text: str, start: int):
    """Parameters of the fn whose `(` is at text[start]: (count without self, has_self)."""
    depth, i, cur, parts = 0, start, "", []
    angle = 0
    while i < len(text):
        ch = text[i]
        if ch in "([{":
            depth += 1
            if depth > 1:
                cur += ch
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                break
            cur += ch
        elif ch == "<":
            angle += 1
            cur += ch
        elif ch == ">" and i > 0 and text[i - 1] != "-":
            angle = max(0, angle - 1)
            cur += ch
        elif ch == "," and depth == 1 and angle == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip():
        parts.append(cur)
    parts = [p.strip() for p in parts if p.strip()]
    has_self = bool(parts) and re.match(r"^(&\s*('\w+\s+)?)?(mut\s+)?self\b", parts[0]) is not None
    return len(parts) - (1 if has_self else 0), has_self


class Crate:
    def __init__(self, src_dir: str):
        self.src = src_dir
        self.modules: dict[tuple, Module] = {}
        self._pending_methods: dict = {}   # (module, type name) -> {method: (vis, n_params, has_self, module)}
        self._load(("crate",), os.path.join(src_dir, "lib.rs"))

    # ------------------------------------------------------------------ loading
    def _load(self, path: tuple, file: str):
        raw = open(file).read()
        text = strip_comments(raw)
        mod = Module(path, file, text=text)
        self.modules[path] = mod
        lines = text.split("\n")
        i = 0
        cfg_skip = False
        while i < len(lines):
            ln = lines[i]
            if ln.startswith("use ") or re.match(r"^" + VIS + r"use\s", ln):
                stmt = ln
                while ";" not in stmt:
                    i += 1
                    stmt += " " + lines[i]
                m = re.match(r"^(" + VIS + r")use\s+(.*?);", " ".join(stmt.split()))
                if m:
                    for p, alias in expand_use(m.group(2)):
                        mod.uses.append((m.group(1).strip(), p, alias))
                i += 1
                continue
            m = ITEM_RE.match(ln)
            if m:
                vis, kind, name = m.group(1).strip(), m.group(2), m.group(3)
                it = Item(kind, name, vis, path)
                if kind == "mod":
                    if ln.rstrip().endswith(";"):
                        self._load_child(mod, path, name)
                    mod.items.setdefault(name, it)
                    i += 1
                    continue
                if kind in ("struct", "enum", "trait", "union"):
                    head = ln
                    j = i
                    # declaration head may span lines (where clauses); find the body opener or the terminating ';'
                    while "{" not in head and ";" not in head and j + 1 < len(lines):
                        j += 1
                        head += " " + lines[j]
                    if kind == "struct" and re.search(r"\bstruct\s+\w+\s*(<[^>]*>)?\s*\(", head):
                        it.tuple_struct = True
                    if "{" in head and not head.rstrip().endswith("}"):
                        j += 1
                        while j < len(lines) and not lines[j].startswith("}"):
                            body = lines[j]
                            if re.match(r"^    \S", body):
                                if kind in ("struct", "union"):
                                    fm = FIELD_RE.match(body)
                                    if fm:
                                        it.fields[fm.group(2)] = fm.group(1).strip()
                                elif kind == "enum":
                                    if not body.strip().startswith("#"):
                                        vm = VARIANT_RE.match(body)
                                        if vm:
                                            it.variants.append(vm.group(1))
                                elif kind == "trait":
                                    fm = MEMBER_FN_RE.match(body)
                                    if fm:
                                        blob = "\n".join(lines[j:j + 40])
                                        n, has_self = count_params(blob, fn_paren(blob))
                                        it.methods[fm.group(2)] = ("pub", n, has_self)
                            j += 1
                    elif "{" in head and kind == "enum":
                        # one-line enum
                        inner = head[head.index("{") + 1: head.rindex("}")]
                        it.variants = [v.strip().split("(")[0].split("=")[0].strip() for v in inner.split(",") if v.strip()]
                    mod.items[name] = it
                    i = max(i + 1, j)
                    continue
                mod.items[name] = it
                if kind == "fn":
                    blob = "\n".join(lines[i:i + 40])
                    n, has_self = count_params(blob, fn_paren(blob))
                    it.methods["()"] = (vis, n, has_self)
                i += 1
                continue
            if ln.startswith('unsafe extern "C" {') or ln.startswith('extern "C" {'):
                j = i + 1
                while j < len(lines) and not lines[j].startswith("}"):
                    fm = MEMBER_FN_RE.match(lines[j])
                    if fm:
                        it = Item("fn", fm.group(2), fm.group(1).strip(), path)
                        blob = lines[j]
                        n, _ = count_params(blob, fn_paren(blob))
                        it.methods["()"] = (it.vis, n, False)
                        mod.items[fm.group(2)] = it
                    j += 1
                i = j
                continue
            m = IMPL_RE.match(ln)
            if m and (ln.startswith("impl") or ln.startswith("unsafe impl")):
                trait, target = m.group(1), m.group(2).split("::")[-1]
                j = i
                head = ln
                while "{" not in head and j + 1 < len(lines):
                    j += 1
                    head += " " + lines[j]
                if head.rstrip().endswith("}"):
                    i = j + 1
                    continue
                j += 1
                while j < len(lines) and not lines[j].startswith("}"):
                    fm = MEMBER_FN_RE.match(lines[j]) if re.match(r"^    \S", lines[j]) else None
                    if fm:
                        blob = "\n".join(lines[j:j + 40])
                        n, has_self = count_params(blob, fn_paren(blob))
                        vis = "pub" if trait else fm.group(1).strip()
                        self._pending_methods.setdefault((path, target), {})[fm.group(2)] = (vis, n, has_self, path)
                    j += 1
                i = j
                continue
            i += 1

    def _load_child(self, parent: Module, path: tuple, name: str):
        base = os.path.dirname(parent.file)
        stem = os.path.splitext(os.path.basename(parent.file))[0]
        cands = []
        if stem in ("lib", "mod", "main"):
            cands = [os.path.join(base, name + ".rs"), os.path.join(base, name, "mod.rs")]
        else:
            cands = [os.path.join(base, stem, name + ".rs"), os.path.join(base, stem, name, "mod.rs")]
        for c in cands:
            if os.path.exists(c):
                self._load(path + (name,), c)
                return

    # ------------------------------------------------------------------ queries
    @staticmethod
    def visible(vis: str, def_mod: tuple, from_mod: tuple) -> bool:
        if vis in ("pub", "pub(crate)"):
            return True
        if vis == "pub(super)":
            scope = def_mod[:-1]
        elif vis.startswith("pub(in "):
            scope = tuple(vis[7:-1].split("::"))
        else:  # private, pub(self)
            scope = def_mod
        return from_mod[: len(scope)] == scope

    def resolve(self, path: tuple, from_mod: tuple, _depth=0):
        """Resolve an item path written in module `from_mod` ("crate::a::B", "super::B", "self::b::C", "B").
        -> (Item or Module, None) on success, (None, reason) when it does not exist or cannot be seen, ("imported-external", None)
        for a name a `use` brought in from another crate, ("external", None) when the first segment is not known to the crate at all
        (the caller decides whether it is one of the crate's dependencies)."""
        if _depth > 12:
            return None, "re-export cycle"
        segs = list(path)
        if segs[0] == "crate":
            cur, segs = ("crate",), segs[1:]
        elif segs[0] in ("super", "self"):
            cur = from_mod
            while segs and segs[0] in ("super", "self"):
                if segs[0] == "super":
                    cur = cur[:-1]
                segs = segs[1:]
        else:
            # a name in scope of from_mod: local item, a `use`, or an external crate
            hit = self._lookup_in_scope(segs[0], from_mod, from_mod, _depth)
            if hit is None:
                return "external", None
            target, why = hit
            if target is None:
                return None, why
            if target in ("external", "imported-external"):
                return target, None
            if len(segs) == 1:
                return target, None
            if isinstance(target, Module):
                cur, segs = target.path, segs[1:]
            else:
                return self._member_path(target, segs[1:], from_mod)
        while segs:
            mod = self.modules.get(cur)
            if mod is None:
                return None, f"no module {'::'.join(cur)}"
            name = segs[0]
            hit = self._lookup_in_scope(name, cur, from_mod, _depth, need_visible=True)
            if hit is None:
                return None, f"`{name}` is not declared in {'::'.join(cur)}"
            target, why = hit
            if target is None:
                return None, why
            segs = segs[1:]
            if not segs or target in ("external", "imported-external"):
                return target, None
            if isinstance(target, Module):
                cur = target.path
            else:
                return self._member_path(target, segs, from_mod)
        return self.modules[cur], None

    def _member_path(self, item: Item, segs, from_mod):
        name = segs[0]
        if item.kind == "enum" and name in item.variants:
            return item, None
        meth = self.methods_of(item).get(name)
        if meth is not None:
            if not self.visible(meth[0], meth[3], from_mod):
                return None, f"{item.name}::{name} is `{meth[0] or 'private'}` in {'::'.join(meth[3])}"
            return item, None
        if item.kind in ("trait", "type"):
            return item, None
        return None, f"{item.name} has no variant or associated item `{name}`"

    def _lookup_in_scope(self, name, mod_path, from_mod, depth, need_visible=False):
        """name declared in or imported into module mod_path.  None: not there.  (target, None) or (None, reason)."""
        mod = self.modules.get(mod_path)
        if mod is None:
            return None
        it = mod.items.get(name)
        if it is not None:
            if need_visible and not self.visible(it.vis, mod_path, from_mod):
                return None, f"`{name}` is `{it.vis or 'private'}` in {'::'.join(mod_path)} (used from {'::'.join(from_mod)})"
            if it.kind == "mod":
                child = self.modules.get(mod_path + (name,))
                return (child if child is not None else it), None
            return it, None
        globs = []
        for vis, p, alias in mod.uses:
            if need_visible and not self.visible(vis, mod_path, from_mod):
                continue
            if p[-1] == "*":
                globs.append(p[:-1])
                continue
            if (alias or p[-1]) == name:
                if p[0] not in ("crate", "super", "self") and self._lookup_in_scope(p[0], mod_path, mod_path, depth) is None:
                    return "imported-external", None
                return self.resolve(p, mod_path, depth + 1)
        for g in globs:
            if g[0] not in ("crate", "super", "self") and self._lookup_in_scope(g[0], mod_path, mod_path, depth) is None:
                continue
            target, _ = self.resolve(g, mod_path, depth + 1)
            if isinstance(target, Module):
                hit = self._lookup_in_scope(name, target.path, mod_path, depth + 1, need_visible=True)
                if hit is not None and hit[0] is not None:
                    return hit
            elif isinstance(target, Item) and target.kind == "enum" and name in target.variants:
                return target, None
        return None

    def methods_of(self, item: Item) -> dict:
        out = {}
        for (mod_path, target), meths in self._pending_methods.items():
            if target == item.name:
                out.update(meths)
        if item.kind == "trait":
            for n, (vis, k, hs) in item.methods.items():
                out.setdefault(n, (vis, k, hs, item.module))
        return out

    def all_member_names(self, from_mod: tuple):
        """Every field and method name of the crate that code in `from_mod` is allowed to touch."""
        names = set()
        for mod in self.modules.values():
            for it in mod.items.values():
                for f, vis in it.fields.items():
                    if self.visible(vis, it.module, from_mod):
                        names.add(f)
                if it.kind == "trait":
                    names.update(it.methods)
        for (mod_path, _t), meths in self._pending_methods.items():
            for n, (vis, _k, _hs, where) in meths.items():
                if self.visible(vis, where, from_mod):
                    names.add(n)
        return names

    def find_type(self, name: str):
        return [it for mod in self.modules.values() for it in mod.items.values() if it.name == name and it.kind in ("struct", "enum", "trait", "type", "union")]
