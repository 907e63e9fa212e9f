"""The boundary documents are held against the header.

* Every ``ctypes.Structure`` INTEGRATION.md shows (the stub a maintainer of the reference copies) is exec'd out of the
  markdown and compared field by field -- name, order, ctype, ``sizeof`` -- with the binding the product uses
  (``morl-baselines_amd/native.py``).
* The product binding itself is compared with what a C compiler makes of ``include/morl_hip.h``: a generated C program
  prints ``sizeof`` / ``offsetof`` of every struct the header defines (gcc, host only; no device code is involved).

A stub that stops short of the header's last field hands the library a struct the library reads past the end of
(round 2: ``morl_update_cfg.rows_total``); this test keeps that from coming back.
"""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = os.path.join(ROOT, "INTEGRATION.md")
HEADER = os.path.join(ROOT, "include", "morl_hip.h")

# ctypes class name <-> header struct, for every struct that crosses the boundary
STRUCTS = {"NetDesc": "morl_net_desc", "UpdateCfg": "morl_update_cfg", "UpdateOut": "morl_update_out", "StepIO": "morl_step_io",
           "ACDesc": "morl_ac_desc", "ACCfg": "morl_ac_cfg", "ACState": "morl_ac_state", "ACBatch": "morl_ac_batch",
           "ACOut": "morl_ac_out", "GPIDesc": "morl_gpi_desc", "GPICfg": "morl_gpi_cfg", "GPIOut": "morl_gpi_out", "GPIBatch": "morl_gpi_batch", "GPIPer": "morl_gpi_per",
           "EnsDesc": "morl_ens_desc", "EnsCfg": "morl_ens_cfg"}


def doc_structs():
    """``class X(C.Structure)`` statements of the markdown's python blocks, exec'd in an empty namespace."""
    text = open(DOC).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    ns = {"C": C}
    found = {}
    for blk in blocks:
        lines = blk.split("\n")
        i = 0
        while i < len(lines):
            m = re.match(r"class (\w+)\(C\.Structure\):", lines[i])
            if not m:
                i += 1
                continue
            j = i + 1
            while j < len(lines) and (lines[j].startswith(" ") or not lines[j].strip()):
                j += 1
            src = "\n".join(lines[i:j])
            exec(compile(src, f"INTEGRATION.md::{m.group(1)}", "exec"), ns)
            found[m.group(1)] = ns[m.group(1)]
            i = j
    return found


def test_integration_md_structs_match_the_binding():
    import morl_baselines_amd.native as native
    found = doc_structs()
    assert {"NetDesc", "UpdateCfg", "UpdateOut", "ACDesc"} <= set(found), sorted(found)
    for name, cls in found.items():
        ref = getattr(native, name)
        got = [(n, t) for n, t in cls._fields_]
        want = [(n, t) for n, t in ref._fields_]
        assert [n for n, _ in got] == [n for n, _ in want], f"{name}: field names / order differ from native.py"
        for (n, t), (_, u) in zip(got, want):
            assert C.sizeof(t) == C.sizeof(u) and getattr(cls, n).offset == getattr(ref, n).offset, f"{name}.{n}"
        assert C.sizeof(cls) == C.sizeof(ref), f"{name}: sizeof {C.sizeof(cls)} != {C.sizeof(ref)}"


def header_struct_fields():
    """{struct: [field, ...]} of every ``typedef struct X { ... } X;`` of the header (comments stripped)."""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} (\w+);", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fp = re.match(r".*\(\s*\*\s*(\w+)\s*\)\s*\(.*\)$", decl, flags=re.S)     # function pointer member
            if fp:
                fields.append(fp.group(1))
                continue
            # "type a, b, *c" / "type d[8]"
            decl = re.sub(r"\[[^\]]*\]", "", decl)
            first, *rest = decl.split(",")
            names = [first.split()[-1]] + [r.strip() for r in rest]
            for n in names:
                fields.append(re.sub(r"\[.*\]", "", n).lstrip("*").strip())
        out[m.group(1)] = fields
    return out


def test_binding_matches_what_a_c_compiler_sees_in_the_header(tmp_path):
    import morl_baselines_amd.native as native
    hdr = header_struct_fields()
    assert set(STRUCTS.values()) <= set(hdr), sorted(set(STRUCTS.values()) - set(hdr))
    assert set(hdr) == set(STRUCTS.values()), f"header structs without a ctypes class: {sorted(set(hdr) - set(STRUCTS.values()))}"
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "morl_hip.h"', "int main(void) {"]
    for cname in STRUCTS.values():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for f in hdr[cname]:
            lines.append(f'  printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    seen = dict(ln.split() for ln in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for pyname, cname in STRUCTS.items():
        cls = getattr(native, pyname)
        assert [n for n, _ in cls._fields_] == hdr[cname], f"{pyname}: field order differs from {cname}"
        assert C.sizeof(cls) == int(seen[cname]), f"sizeof({cname}) = {seen[cname]}, ctypes {C.sizeof(cls)}"
        for n, _ in cls._fields_:
            assert getattr(cls, n).offset == int(seen[f"{cname}.{n}"]), f"{cname}.{n}"


def test_every_exported_symbol_is_mapped_to_a_reference_call_site():
    """Section 3 of INTEGRATION.md names every symbol of the header (or covers it with a documented prefix / wildcard)."""
    from morl_baselines_amd.native import EXPORTED_SYMBOLS
    text = open(DOC).read()
    missing = [s for s in EXPORTED_SYMBOLS if not re.search(r"`" + s + r"`", text)
               and not re.search(r"`" + s.rsplit("_", 1)[0] + r"` / `_" + s.rsplit("_", 1)[1] + r"`", text)
               and not re.search(r"`_" + s.rsplit("_", 1)[1] + r"`", text)]
    assert not missing, f"INTEGRATION.md section 3 does not mention: {missing}"


def test_environment_switches_are_one_list():
    """every ``MORL_*`` environment variable the sources read is in ``native.KNOWN_ENV`` and in INTEGRATION.md's
    table, nothing else is, and an unknown ``MORL_*`` variable is refused loudly instead of being ignored."""
    sys.path.insert(0, ROOT)
    import morl_baselines_amd.native as native
    read = set()
    pat = re.compile(r"""(?:getenv\(\s*|environ\.get\(\s*|environ\.setdefault\(\s*|environ\[\s*)["'](MORL_[A-Z0-9_]+)["']""")
    files = [os.path.join(ROOT, f) for f in ("bench.py", "bench_ac.py", "bench_front.py", "__graft_entry__.py", "oracle/ref_harness.py")]
    for top in ("morl-baselines_amd", "tools"):
        for d, _, fs in os.walk(os.path.join(ROOT, top)):
            files += [os.path.join(d, f) for f in fs if f.endswith((".py", ".hip", ".h"))]
    for f in files:
        read |= set(pat.findall(open(f, errors="ignore").read()))
    assert read == set(native.KNOWN_ENV), (sorted(read - set(native.KNOWN_ENV)), sorted(set(native.KNOWN_ENV) - read))
    doc = open(DOC).read()
    section = doc[doc.index("## Run-time switches"):]
    for name in native.KNOWN_ENV:
        assert f"`{name}`" in section or f"`{name}=" in section, f"{name} is not in INTEGRATION.md's switch table"
    native.check_environment({"MORL_EXACT_F32": "1", "HOME": "/"})
    with pytest.raises(RuntimeError, match="MORL_EXACT_F23"):
        native.check_environment({"MORL_EXACT_F23": "1"})
