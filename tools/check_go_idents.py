#!/usr/bin/env python3
"""Resolve every package-qualified identifier of the Go shim (go/**) without a Go toolchain (there is none in the build image).

For every `alias.Ident` in a .go file whose alias is an import of
  * github.com/consensys/gnark/<path>     -> the identifier must be DECLARED (func / type / const / var, exported) in a non-test
                                             .go file of /root/reference/<path>, or of go/<path> for the shim's own packages;
  * github.com/consensys/gnark-crypto/... -> the module is not in /root/reference (go.mod dependency): the same `pkg.Ident`
                                             must be USED somewhere in the reference with the same import path (gnark itself
                                             compiles against it, so the name exists in the pinned version);
  * the standard library                   -> checked against a small allow-list of the packages the shim uses;
  * "C"                                    -> `C.ga_*` must be a prototype of include/gnark_amd.h with the SAME NUMBER of
                                             arguments at the call site; `C.GA_*` must be a #define of the header; a handful of
                                             cgo built-ins (C.int, C.GoString, ...) are allowed.
Methods and struct fields (x.Method() on values) are out of reach of a textual check and are not claimed.

  python tools/check_go_idents.py            # prints a report, exit code 1 on any unresolved identifier
  python tools/check_go_idents.py --json go/IDENTS.json   # also writes the resolved table (committed, reviewed by the test)
"""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GNARK = "github.com/consensys/gnark/"
CRYPTO = "github.com/consensys/gnark-crypto/"
SHIM_PREFIX = "backend/accelerated/mi355x"

STDLIB = {
    "fmt": {"Errorf", "Sprintf", "Println", "Printf", "Sprint", "Fprintf"}, "errors": {"New", "Is", "As"},
    "os": {"Open", "File", "Getenv", "Create", "Remove", "CreateTemp", "ReadFile", "WriteFile", "MkdirAll"}, "slices": {"Equal", "Clone", "Concat", "Sort", "Compact", "Contains"},
    "sync": {"Mutex", "RWMutex", "Once", "WaitGroup"}, "time": {"Now", "Since", "Duration"}, "unsafe": {"Pointer", "Sizeof", "SliceData", "Slice", "Add"},
    "runtime": {"Pinner", "KeepAlive", "LockOSThread", "UnlockOSThread", "NumCPU"}, "math/big": {"Int", "NewInt"}, "io": {"Reader", "Writer", "ReaderFrom", "WriterTo"},
    "bytes": {"Buffer", "NewReader", "Equal"}, "testing": {"T", "B", "Short"}, "math/bits": {"Len64", "TrailingZeros64"}, "hash": {"Hash"},
    "context": {"Context", "Background"}, "strings": {"Join", "Split", "HasPrefix"}, "sort": {"Slice", "Ints"}, "path/filepath": {"Join"},
    "encoding/json": {"MarshalIndent", "Marshal"},
    "crypto/sha256": {"New", "Sum256"}, "encoding/binary": {"BigEndian", "LittleEndian", "Write", "Read"},
}
CGO_BUILTINS = {"int", "uint", "uint32_t", "uint64_t", "int32_t", "size_t", "char", "uchar", "GoString", "CString", "free", "malloc", "calloc", "GoBytes",
                "uint8_t", "uintptr_t", "longlong", "ulonglong", "ulong", "long", "double", "float", "uint16_t", "int64_t", "schar", "short", "ushort"}


def go_files(top):
    for d, _, fs in os.walk(top):
        for f in fs:
            if f.endswith(".go"):
                yield os.path.join(d, f)


def strip_comments_and_strings(src):
    src = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r'`[^`]*`', '""', src)
    src = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', src)
    return src


def imports_of(src):
    """alias -> import path"""
    out = {}
    block = re.search(r"^import\s*\((.*?)^\)", src, flags=re.S | re.M)
    lines = block.group(1).splitlines() if block else []
    lines += re.findall(r'^import\s+((?:\w+\s+)?"[^"]+")', src, flags=re.M)
    for ln in lines:
        m = re.match(r'\s*(?:(\w+|\.|_)\s+)?"([^"]+)"', ln)
        if m:
            path = m.group(2)
            alias = m.group(1) or default_alias(path)
            out[alias] = path
    return out


_PKG_NAME_CACHE = {}


def default_alias(path):
    """package name of an import path: the `package x` clause when the directory is known (reference / shim), else the last element"""
    if path in _PKG_NAME_CACHE:
        return _PKG_NAME_CACHE[path]
    name = path.rsplit("/", 1)[-1]
    d = None
    if path.startswith(GNARK):
        rel = path[len(GNARK):]
        for base in (os.path.join(ROOT, "go"), REF):
            if os.path.isdir(os.path.join(base, rel)):
                d = os.path.join(base, rel)
                break
    if d:
        for f in sorted(os.listdir(d)):
            if f.endswith(".go") and not f.endswith("_test.go"):
                m = re.search(r"^package\s+(\w+)", open(os.path.join(d, f)).read(), flags=re.M)
                if m:
                    name = m.group(1)
                    break
    _PKG_NAME_CACHE[path] = name
    return name


_DECL_CACHE = {}


def declared_in(dirpath):
    """exported top-level identifiers declared in the non-test .go files of a package directory"""
    if dirpath in _DECL_CACHE:
        return _DECL_CACHE[dirpath]
    names = set()
    if os.path.isdir(dirpath):
        for f in os.listdir(dirpath):
            if not f.endswith(".go") or f.endswith("_test.go"):
                continue
            src = strip_comments_and_strings(open(os.path.join(dirpath, f)).read())
            names |= set(re.findall(r"^func\s+([A-Z]\w*)\s*[\(\[]", src, flags=re.M))
            names |= set(re.findall(r"^type\s+([A-Z]\w*)\b", src, flags=re.M))
            names |= set(re.findall(r"^(?:var|const)\s+([A-Z]\w*)\b", src, flags=re.M))
            for blk in re.findall(r"^(?:var|const|type)\s*\((.*?)^\)", src, flags=re.S | re.M):
                names |= set(re.findall(r"^\s*([A-Z]\w*)\b", blk, flags=re.M))
    _DECL_CACHE[dirpath] = names
    return names


def declared_by_patches():
    """exported identifiers ADDED to a reference package by a patch shipped under go/** (prove.patch: the PLONK Accelerator hooks;
    internal/fixtures/groth16_rs_*.patch: the test-only TestingHooks of the Groth16 prover): package dir -> {ident: patch}"""
    out = {}
    for d, _, fs in os.walk(os.path.join(ROOT, "go")):
        for f in fs:
            if not f.endswith(".patch"):
                continue
            pkg = None
            for line in open(os.path.join(d, f)):
                if line.startswith("+++ "):
                    m = re.match(r"\+\+\+ [ab]/(\S+)", line)
                    pkg = os.path.dirname(m.group(1)) if m else None
                elif pkg and line.startswith("+") and not line.startswith("+++"):
                    m = re.match(r"\+(?:func|type|var|const)\s+([A-Z]\w*)\b", line)
                    if m:
                        out.setdefault(pkg, {})[m.group(1)] = os.path.relpath(os.path.join(d, f), ROOT)
    return out


_REF_USE_CACHE = {}


def used_in_reference(import_path, ident):
    """is `<alias>.<ident>` used in a reference file that imports import_path (under whatever alias)?"""
    key = import_path
    if key not in _REF_USE_CACHE:
        uses = set()
        for f in go_files(REF):
            try:
                raw = open(f).read()
            except OSError:
                continue
            if '"%s"' % import_path not in raw:
                continue
            src = strip_comments_and_strings_keep_imports(raw)
            for alias, path in imports_of(raw).items():
                if path == import_path:
                    uses |= set(re.findall(r"\b%s\.([A-Z]\w*)" % re.escape(alias), src))
        _REF_USE_CACHE[key] = uses
    return ident in _REF_USE_CACHE[key]


def strip_comments_and_strings_keep_imports(raw):
    return strip_comments_and_strings(raw)


def header_symbols():
    h = open(os.path.join(ROOT, "include", "gnark_amd.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(ga_\w+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    defines = set(re.findall(r"^#define\s+(GA_\w+)", h, flags=re.M))
    types = set(re.findall(r"\btypedef\s+struct\s+\w*\s*(?:\{[^}]*\})?\s*(ga_\w+)\s*;", h, flags=re.S)) | set(re.findall(r"\}\s*(ga_\w+)\s*;", h))
    return protos, defines, types


def call_arity(src, pos):
    """number of top-level arguments of the call whose '(' is at src[pos]"""
    depth, n, i, seen = 0, 0, pos, False
    while i < len(src):
        ch = src[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                return n + (1 if seen else 0)
        elif ch == "," and depth == 1:
            n += 1
        elif depth >= 1 and not ch.isspace():
            seen = True
        i += 1
    return -1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        print("reference tree not present at %s: nothing to resolve against" % REF)
        return 2
    protos, defines, ctypes = header_symbols()
    patched = declared_by_patches()
    table, bad = [], []
    for f in sorted(go_files(os.path.join(ROOT, "go"))):
        raw = open(f).read()
        imps = imports_of(raw)
        src = strip_comments_and_strings(raw)
        rel = os.path.relpath(f, ROOT)
        has_c = re.search(r'^import\s+"C"', raw, flags=re.M) is not None
        if has_c:
            for m in re.finditer(r"\bC\.(\w+)", src):
                name = m.group(1)
                if name.startswith("ga_") and name in protos:
                    after = src[m.end():m.end() + 1]
                    if after == "(":
                        got = call_arity(src, m.end())
                        if got != protos[name]:
                            bad.append((rel, "C." + name, "called with %d arguments, header declares %d" % (got, protos[name])))
                            continue
                    table.append((rel, "C." + name, "include/gnark_amd.h prototype (%d args)" % protos[name]))
                elif name.startswith("GA_") and name in defines:
                    table.append((rel, "C." + name, "include/gnark_amd.h #define"))
                elif name in ctypes or name.startswith("struct_") or name in CGO_BUILTINS:
                    table.append((rel, "C." + name, "cgo / header type"))
                else:
                    bad.append((rel, "C." + name, "not in include/gnark_amd.h"))
        for alias, path in imps.items():
            if alias in ("_", ".", "C"):
                continue
            idents = set(re.findall(r"(?<![\w.])%s\.([A-Za-z_]\w*)" % re.escape(alias), src))
            for ident in sorted(idents):
                if path.startswith(GNARK):
                    pkg = path[len(GNARK):]
                    d = os.path.join(ROOT, "go", pkg) if pkg.startswith(SHIM_PREFIX) else os.path.join(REF, pkg)
                    if ident in declared_in(d):
                        table.append((rel, "%s.%s" % (alias, ident), "declared in %s" % os.path.relpath(d, "/")))
                    elif ident in patched.get(pkg, {}):
                        table.append((rel, "%s.%s" % (alias, ident), "added to %s by %s" % (pkg, patched[pkg][ident])))
                    else:
                        bad.append((rel, "%s.%s" % (alias, ident), "not declared in %s" % d))
                elif path.startswith(CRYPTO):
                    if used_in_reference(path, ident):
                        table.append((rel, "%s.%s" % (alias, ident), "gnark-crypto [EXT]: used by the reference with import %s" % path))
                    else:
                        bad.append((rel, "%s.%s" % (alias, ident), "gnark-crypto %s: no use of this identifier anywhere in the reference" % path))
                elif path in STDLIB:
                    if ident in STDLIB[path]:
                        table.append((rel, "%s.%s" % (alias, ident), "standard library"))
                    else:
                        bad.append((rel, "%s.%s" % (alias, ident), "not in the allow-list for standard package %s" % path))
                else:
                    bad.append((rel, "%s.%s" % (alias, ident), "import %s is neither gnark, gnark-crypto nor an allow-listed standard package" % path))
    print("%d package-qualified identifiers resolved, %d unresolved" % (len(table), len(bad)))
    for r in bad:
        print("UNRESOLVED  %s: %s -- %s" % r)
    if args.json:
        json.dump({"resolved": sorted(set(table)), "unresolved": bad}, open(args.json, "w"), indent=0)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
