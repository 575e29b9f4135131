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


# ---- prototype-level check (tests/_gocgo.py): argument TYPES against the header, pointer rules -------------------------------

def test_every_cgo_argument_has_the_type_the_header_declares_and_every_pointer_is_safe():
    from tests._gocgo import Checker
    c = Checker().run()
    assert not c.problems, "\n".join(c.problems)
    assert c.n_calls >= 40 and c.n_args >= 180 and c.n_pointers >= 80


def _problems(edit):
    from tests._gocgo import Checker
    return Checker(edit).run().problems


def _sub(fname, old, new, count=1):
    def edit(name, text):
        if name != fname:
            return text
        assert old in text, (fname, old)
        return text.replace(old, new, count)
    return edit


def test_the_checker_catches_each_class_of_mistake():
    """the same files with one mistake planted each: a checker that passes everything proves nothing"""
    # a []float32 handed over as uint32_t*
    p = _problems(_sub("index.go", "C.qv_index_add(d.h, f32p(flat)", "C.qv_index_add(d.h, u32p(flat)"))
    assert any("u32p" in x and "[]float32" in x for x in p) and any("wants *C.float" in x for x in p), p
    # an integer of the wrong C type
    p = _problems(_sub("index.go", "C.qv_index_update(d.h, C.uint32_t(row)", "C.qv_index_update(d.h, C.int(row)"))
    assert any("has type C.int, the header wants C.uint32_t" in x for x in p), p
    # a missing argument
    p = _problems(_sub("sharded.go", "C.qv_sharded_remove(d.h, u32p(rows), C.uint32_t(len(rows)))", "C.qv_sharded_remove(d.h, u32p(rows))"))
    assert any("called with 2 arguments, declared with 3" in x for x in p), p
    # &slice[0] without the emptiness guard
    p = _problems(_sub("metric.go", "\tif n == 0 {\n\t\treturn out, nil\n\t}\n", ""))
    assert sum("empty slice" in x for x in p) == 3, p
    # the helpers' own guard
    p = _problems(_sub("index.go", "func u32p(s []uint32) *C.uint32_t {\n\tif len(s) == 0 {\n\t\treturn nil\n\t}\n", "func u32p(s []uint32) *C.uint32_t {\n"))
    assert any("helper u32p lacks" in x for x in p), p
    # the address of a Go-typed variable as an out parameter
    p = _problems(_sub("index.go", "\tvar n C.uint32_t\n", "\tvar n uint32\n"))
    assert any("non-C variable &n" in x for x in p), p
    # a handle of the wrong kind
    p = _problems(_sub("graph.go", "C.qv_graph_destroy(h.g)", "C.qv_graph_destroy(h.idx)"))
    assert any("has type *C.qv_index, the header wants *C.qv_graph" in x for x in p), p
    # reinterpreting a slice as a pointer to a different element type
    p = _problems(_sub("graph.go", "(*C.int8_t)(unsafe.Pointer(&levels[0]))", "(*C.uint32_t)(unsafe.Pointer(&levels[0]))"))
    assert any("reinterprets []int8 as *C.uint32_t" in x for x in p), p


def test_nothing_in_the_binding_lets_c_keep_a_go_pointer():
    """cgo's rule: C may not keep a Go pointer after the call returns.  include/qv.h promises that no caller pointer is retained;
    on the Go side that leaves only indirect ways to break the rule, and none of them is used: no Go memory reachable from a
    C-allocated struct, no handle or callback, and every slice whose address crosses holds C scalars only (checked above)."""
    for path in GO:
        t = re.sub(r"//[^\n]*", "", open(path).read())
        for banned in ("C.malloc", "C.calloc", "C.CBytes", "C.CString", "cgo.Handle", "cgo.NewHandle", "//export", "uintptr(unsafe.Pointer"):
            assert banned not in t, (os.path.basename(path), banned)
    header = open(os.path.join(ROOT, "include", "qv.h")).read()
    assert "no\n *     caller pointer is retained after a call returns" in header
