"""A prototype-level check of the cgo calls in go/quivergpu/*.go against include/qv.h, for an image without a Go toolchain
(the reference builds with CGO_ENABLED=1, Dockerfile:20).  Not a Go compiler: a small type inferencer for the handful of
expression forms the binding uses as C-call arguments, strict about anything it does not recognise (an unknown form is a test
failure, not a pass).  What it establishes for every C.qv_* call:

  1. each argument's cgo type equals the parameter type the header declares (uint32_t -> C.uint32_t, const float* -> *C.float,
     qv_index** -> **C.qv_index, ...; nil for pointers, untyped constants for integers);
  2. every pointer into Go memory is the address of element 0 of a slice that cannot be empty there (a helper with a
     len == 0 -> nil guard, or a guard / a constant-length make in the calling function), or of a local C-typed variable;
  3. the memory such a pointer reaches holds no Go pointers (slices of C scalars only), and no C call receives a pointer it
     could keep: the header's contract is that nothing is retained after return, so there is no cgo.Handle, no C.malloc'd struct
     holding Go memory, and no callback in the binding.
"""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = sorted(glob.glob(os.path.join(ROOT, "go", "quivergpu", "*.go")))

SCALARS = {"float32": "float", "uint32": "uint32_t", "int8": "int8_t", "uint64": "uint64_t", "int32": "int32_t"}
C_INTS = {"C.int", "C.uint32_t", "C.uint64_t", "C.size_t", "C.int8_t", "C.qv_metric"}


def strip_comments(t):
    t = re.sub(r"//[^\n]*", "", t)
    return re.sub(r"/\*.*?\*/", "", t, flags=re.S)


def header_prototypes():
    """name -> (return cgo type, [param cgo types])"""
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "qv.h")).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z_0-9 \*]*?)\b(qv_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or not ret:
            continue
        params = [] if args in ("", "void") else [cgo_of_c(a) for a in split_top(args)]
        protos[name] = (cgo_of_c(ret + " x"), params)
    return protos


def cgo_of_c(decl):
    """'const float* rows' -> '*C.float';  'qv_index** out' -> '**C.qv_index';  'void* stream' -> 'unsafe.Pointer'"""
    d = re.sub(r"\bconst\b", " ", decl)
    d = re.sub(r"\s+", " ", d).strip()
    stars = d.count("*")
    d = d.replace("*", " ")
    toks = d.split()
    base = toks[0] if len(toks) == 1 else " ".join(toks[:-1])         # drop the parameter name
    if base in ("unsigned", "unsigned int"):
        base = "uint"
    if base == "void":
        return "unsafe.Pointer" if stars == 1 else ("void" if stars == 0 else "*" * (stars - 1) + "unsafe.Pointer")
    return "*" * stars + "C." + base


def split_top(s):
    out, depth, cur = [], 0, ""
    for c in s:
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
        if c == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += c
    if cur.strip():
        out.append(cur.strip())
    return out


class GoFile:
    def __init__(self, path, text=None):
        self.path = path
        self.text = strip_comments(open(path).read() if text is None else text)
        self.funcs = []                                  # (start, end, header, body)
        for m in re.finditer(r"^func [^\n]*\{\s*$|^func [^\n]*\{[^\n]*\}\s*$", self.text, flags=re.M):
            start = m.start()
            i = self.text.index("{", m.start() + 4 + self._sig_end(self.text[m.start():]))
            depth, j = 1, i + 1
            while depth:
                depth += {"{": 1, "}": -1}.get(self.text[j], 0)
                j += 1
            self.funcs.append((start, j, self.text[start:i], self.text[i:j]))

    @staticmethod
    def _sig_end(t):
        """offset of the body's opening brace: the first '{' at parenthesis depth 0 that is not part of 'struct{' / 'interface{' / 'func() {'"""
        depth = 0
        for i, c in enumerate(t):
            if c in "([":
                depth += 1
            elif c in ")]":
                depth -= 1
            elif c == "{" and depth == 0:
                return i - 4
        raise ValueError("no body")

    def enclosing(self, pos):
        best = None
        for f in self.funcs:
            if f[0] <= pos < f[1] and (best is None or f[0] >= best[0]):
                best = f
        return best


def struct_fields(files):
    """(struct name, field) -> Go type, for every struct in the package"""
    out = {}
    for gf in files:
        for m in re.finditer(r"type (\w+) struct \{(.*?)\n\}", gf.text, flags=re.S):
            for line in m.group(2).split("\n"):
                fm = re.match(r"\s*([\w, ]+?)\s+([\*\[\]\w\.]+)\s*$", line)
                if fm:
                    for nm in fm.group(1).split(","):
                        out[(m.group(1), nm.strip())] = fm.group(2)
    return out


def helper_returns(files):
    """name -> (param Go type, return type) of one-parameter helpers returning a C pointer, with the guard checked"""
    out = {}
    for gf in files:
        for m in re.finditer(r"func (\w+)\((\w+) (\[\]\w+)\) (\*C\.\w+) \{(.*?)\n\}", gf.text, flags=re.S):
            name, p, pt, rt, body = m.groups()
            guarded = re.search(r"if len\(%s\) == 0 \{\s*return nil\s*\}" % p, body) is not None
            ok_conv = re.search(r"return \(%s\)\(unsafe\.Pointer\(&%s\[0\]\)\)" % (re.escape(rt), p), body) is not None
            out[name] = (pt, rt, guarded and ok_conv)
    return out


class Checker:
    def __init__(self, edit=None):
        """edit: optional function (basename, text) -> text applied to every source before checking (the tests mutate the binding
        to show that each class of mistake is caught)"""
        self.files = [GoFile(p, edit(os.path.basename(p), open(p).read()) if edit else None) for p in GO]
        self.fields = struct_fields(self.files)
        self.helpers = helper_returns(self.files)
        self.protos = header_prototypes()
        self.problems = []
        self.n_calls = 0
        self.n_args = 0
        self.n_pointers = 0

    # ---- declarations visible in a function ------------------------------------------------------------------------
    def locals_of(self, func):
        head, body = func[2], func[3]
        types = {}
        rm = re.match(r"func \((\w+) \*?(\w+)\)", head)
        if rm:
            types[rm.group(1)] = rm.group(2)                               # receiver: struct name
        pm = re.search(r"\)\s*\w*\((.*)\)\s*(\(.*\)|[\*\[\]\w\.]+)?\s*$", head, flags=re.S) or re.search(r"func \w+\((.*?)\)", head, flags=re.S)
        plist = re.search(r"func (?:\([^)]*\) )?\w+\((.*?)\)(?: |$)", head, flags=re.S)
        if plist:
            pend = []
            for p in split_top(plist.group(1)):
                toks = p.split()
                if len(toks) == 1:
                    pend.append(toks[0])
                elif toks:
                    for nm in pend + [toks[0]]:
                        types[nm] = " ".join(toks[1:])
                    pend = []
        for m in re.finditer(r"\bvar ([\w, ]+?) ([\*\[\]\w\.]+)\s*$", body, flags=re.M):
            for nm in m.group(1).split(","):
                types[nm.strip()] = m.group(2)
        for m in re.finditer(r"^\s*([\w, ]+?) :?= (make\(.*)$", body, flags=re.M):      # a, b := make([]T, n), make([]U, m)
            names = [x.strip() for x in m.group(1).split(",")]
            makes = split_top(m.group(2))
            if len(names) == len(makes):
                for nm, mk in zip(names, makes):
                    mm = re.match(r"make\((\[\][\w\.]+)", mk)
                    if mm:
                        types[nm] = mm.group(1)
        for m in re.finditer(r"(\w+): make\((\[\][\w\.]+), ([^)]*(?:\([^)]*\)[^)]*)*)\)", body):      # composite literal fields
            types["." + m.group(1)] = m.group(2)
        return types

    def type_of(self, expr, func, types):
        e = expr.strip()
        if e == "nil":
            return "nil"
        if re.fullmatch(r"\d+", e):
            return "untyped-int"
        if re.fullmatch(r"C\.QV_[A-Z_0-9]+", e):
            return "untyped-int"
        m = re.fullmatch(r"(C\.\w+)\((.*)\)", e, flags=re.S)
        if m and not m.group(1).startswith("C.qv_") or (m and m.group(1) in ("C.qv_metric",)):
            return m.group(1)
        m = re.fullmatch(r"\((\*C\.\w+)\)\(unsafe\.Pointer\(&([\w\.]+)\[0\]\)\)", e)
        if m:
            self.pointer_into_slice(m.group(2), m.group(1), func, types, e)
            return m.group(1)
        m = re.fullmatch(r"(\w+)\((.+)\)", e, flags=re.S)
        if m and m.group(1) in self.helpers:
            pt, rt, ok = self.helpers[m.group(1)]
            if not ok:
                self.problems.append("helper %s lacks the len == 0 -> nil guard" % m.group(1))
            at = self.slice_type(m.group(2).strip(), func, types)
            if at is not None and at != pt:
                self.problems.append("%s: helper %s takes %s, got %s (%s)" % (os.path.basename(self.cur.path), m.group(1), pt, at, e))
            if at is None:
                self.problems.append("%s: cannot type the argument of helper call %s" % (os.path.basename(self.cur.path), e))
            self.n_pointers += 1
            return rt
        m = re.fullmatch(r"&\[\](C\.\w+)\{[^}]*\}\[0\]", e)
        if m:
            self.n_pointers += 1
            return "*" + m.group(1)
        m = re.fullmatch(r"&(\w+)\[0\]", e)
        if m:
            st = types.get(m.group(1))
            if not st or not st.startswith("[]"):
                self.problems.append("cannot type %s" % e); return "?"
            self.pointer_into_slice(m.group(1), "*" + st[2:], func, types, e)
            return "*" + st[2:]
        m = re.fullmatch(r"&(\w+)", e)
        if m:
            t = types.get(m.group(1))
            if t is None:
                self.problems.append("cannot type %s" % e); return "?"
            if not (t.startswith("C.") or t.startswith("*C.")):
                self.problems.append("%s: address of a non-C variable %s passed to C" % (os.path.basename(self.cur.path), e))
            self.n_pointers += 1
            return "*" + t
        m = re.fullmatch(r"(\w+)\.(\w+)", e)
        if m and m.group(1) in types:
            ft = self.fields.get((types[m.group(1)].lstrip("*"), m.group(2)))
            if ft:
                return ft
        if re.fullmatch(r"\w+", e) and e in types:
            return types[e]
        self.problems.append("%s: unrecognised argument form %r" % (os.path.basename(self.cur.path), e))
        return "?"

    def slice_type(self, e, func, types):
        if re.fullmatch(r"\w+", e):
            t = types.get(e)
            if t == "vectortypes.F32":
                return "[]float32"
            return t
        m = re.fullmatch(r"(\w+)\.(\w+)", e)
        if m:
            base = types.get(m.group(1), "").lstrip("*")
            return self.fields.get((base, m.group(2))) or types.get("." + m.group(2))
        return None

    def pointer_into_slice(self, sl, ptr_type, func, types, expr):
        """&sl[0] converted to ptr_type: element type must be the C scalar's Go twin, and sl must be non-empty at this point"""
        self.n_pointers += 1
        st = self.slice_type(sl, func, types)
        where = os.path.basename(self.cur.path)
        if st is None or not st.startswith("[]"):
            self.problems.append("%s: cannot find the slice type of %s in %s" % (where, sl, expr)); return
        el = st[2:]
        want = ptr_type[1:]
        if not (el == want or (el in SCALARS and "C." + SCALARS[el] == want)):
            self.problems.append("%s: %s reinterprets []%s as %s" % (where, expr, el, ptr_type))
        if not self.non_empty(sl, func):
            self.problems.append("%s: %s may take the address of element 0 of an empty slice" % (where, expr))

    def non_empty(self, sl, func):
        body = func[3][: self.cur_call_off]
        name = re.escape(sl)
        field = re.escape(sl.split(".")[-1])
        if re.search(r"if len\(%s\) > 0 \{" % name, body) or re.search(r"if len\(%s\) == 0 \{\s*return" % name, body):
            return True
        if re.search(r"\b%s :?= make\(\[\][\w\.]+, (\d+|[^)]*\+ ?1)\)" % name, body):        # constant or x+1 elements
            return True
        # a variable tied to the slice's length and checked against 0 before the call
        tied = set()
        for m in re.finditer(r"(\w+) := len\(%s\)(?: / \w+)?\s*$" % name, body, flags=re.M):
            tied.add(m.group(1))
        for m in re.finditer(r"(?:\b%s :?= |\b%s: )make\(\[\][\w\.]+, (\w+)\)" % (name, field), body):
            tied.add(m.group(1))
        # equal lengths enforced: len(a) != len(b) -> return
        for m in re.finditer(r"len\((\w+)\) != len\(%s\)|len\(%s\) != len\((\w+)\)" % (name, name), body):
            other = m.group(1) or m.group(2)
            for m2 in re.finditer(r"(\w+) := len\(%s\)(?: / \w+)?\s*$" % re.escape(other), body, flags=re.M):
                tied.add(m2.group(1))
        return any(re.search(r"if %s == 0 \{\s*return" % re.escape(v), body) for v in tied)

    def compatible(self, got, want):
        if got == want:
            return True
        if got == "nil":
            return want.startswith("*") or want == "unsafe.Pointer"
        if got == "untyped-int":
            return want in C_INTS
        return False

    def run(self):
        for gf in self.files:
            self.cur = gf
            for m in re.finditer(r"C\.(qv_[a-z_0-9]+)\(", gf.text):
                name = m.group(1)
                if name in ("qv_metric", "qv_status"):
                    continue
                if name not in self.protos:
                    self.problems.append("%s: %s is not declared in include/qv.h" % (os.path.basename(gf.path), name)); continue
                i, depth = m.end(), 1
                while depth:
                    depth += {"(": 1, "[": 1, "{": 1, ")": -1, "]": -1, "}": -1}.get(gf.text[i], 0)
                    i += 1
                args = split_top(gf.text[m.end():i - 1])
                ret, params = self.protos[name]
                self.n_calls += 1
                if len(args) != len(params):
                    self.problems.append("%s: %s called with %d arguments, declared with %d" % (os.path.basename(gf.path), name, len(args), len(params))); continue
                func = gf.enclosing(m.start())
                if func is None:
                    self.problems.append("%s: call of %s outside any function" % (os.path.basename(gf.path), name)); continue
                # nested function literals: declarations of the outermost enclosing function are visible too
                outer = min((f for f in gf.funcs if f[0] <= m.start() < f[1]), key=lambda f: f[0])
                types = self.locals_of(outer)
                types.update(self.locals_of(func))
                self.cur_call_off = m.start() - outer[0] - len(outer[2])
                for a, p in zip(args, params):
                    self.n_args += 1
                    t = self.type_of(a, outer, types)
                    if t != "?" and not self.compatible(t, p):
                        self.problems.append("%s: %s: argument %r has type %s, the header wants %s" % (os.path.basename(gf.path), name, a, t, p))
        return self
