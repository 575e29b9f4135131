"""The Go side of the boundary ships as files (go/quivergpu/*.go) that this image cannot compile (no Go toolchain): check them
against include/qv.h instead — every C.qv_* call names a function the header declares and passes the declared number of
arguments, every C.QV_* constant exists, and the files implement core.Index + core.BatchIndex (pkg/core/collection.go:78-96)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = sorted(glob.glob(os.path.join(ROOT, "go", "quivergpu", "*.go")))


def _header():
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "qv.h")).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(qv_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    consts = set(re.findall(r"#define\s+(QV_[A-Z_0-9]+)", src)) | set(re.findall(r"\b(QV_[A-Z_0-9]+)\s*=", src))
    return protos, consts


def _calls(text):
    """(name, argc) of every C.qv_*( ... ) call, by bracket matching"""
    out = []
    for m in re.finditer(r"C\.(qv_[a-z_0-9]+)\(", text):
        if m.group(1) in ("qv_metric", "qv_status"):               # a type conversion, not a call
            continue
        i, depth, argc, seen = m.end(), 1, 0, False
        while depth:
            c = text[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            elif c == "," and depth == 1:
                argc += 1
            elif not c.isspace():
                seen = True
            i += 1
        out.append((m.group(1), argc + 1 if seen else 0))
    return out


def test_go_files_exist():
    names = {os.path.basename(p) for p in GO}
    assert {"index.go", "sharded.go", "graph.go", "metric.go", "doc.go"} <= names


def test_every_cgo_call_matches_the_header():
    protos, consts = _header()
    n_calls = 0
    for path in GO:
        text = re.sub(r"//[^\n]*", "", open(path).read())
        for name, argc in _calls(text):
            assert name in protos, "%s: %s is not declared in include/qv.h" % (os.path.basename(path), name)
            assert argc == protos[name], "%s: %s called with %d arguments, declared with %d" % (os.path.basename(path), name, argc, protos[name])
            n_calls += 1
        for c in re.findall(r"C\.(QV_[A-Z_0-9]+)", text):
            assert c in consts, "%s: C.%s is not in include/qv.h" % (os.path.basename(path), c)
        for t in re.findall(r"\*C\.(qv_[a-z_]+)\b(?!\()", text):
            assert t in ("qv_index", "qv_graph", "qv_sharded"), t
    assert n_calls >= 35


def test_core_index_and_batch_index_are_implemented():
    text = open(os.path.join(ROOT, "go", "quivergpu", "index.go")).read()
    for sig in (r"func \(x \*Index\) Insert\(id string, v vectortypes\.F32\) error",
                r"func \(x \*Index\) Delete\(id string\) error",
                r"func \(x \*Index\) Search\(q vectortypes\.F32, k int\) \(\[\]types\.BasicSearchResult, error\)",
                r"func \(x \*Index\) Size\(\) int",
                r"func \(x \*Index\) InsertBatch\(vs map\[string\]vectortypes\.F32\) error",
                r"func \(x \*Index\) DeleteBatch\(ids \[\]string\) error"):
        assert re.search(sig, text), sig
    for path in GO:                                               # balanced braces / parentheses: a cheap syntax check
        t = re.sub(r"//[^\n]*", "", open(path).read())
        t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
        t = re.sub(r'"(\\.|[^"\\])*"', '""', t)
        t = re.sub(r"`[^`]*`", "``", t)
        for a, b in ("()", "{}", "[]"):
            assert t.count(a) == t.count(b), (os.path.basename(path), a)
