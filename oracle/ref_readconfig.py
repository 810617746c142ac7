"""TEST INFRASTRUCTURE - ctypes view of the REFERENCE's own config parser.

oracle/_ref/libreadconfig_ref.so is /root/reference/src/tools/readconfig.c compiled where it lies
(oracle/Makefile); nothing here restates it.  Only tests/ and the golden-fixture generator import this.
`dump(...)` returns everything the product's parser is compared on, as plain JSON-able data.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class RcData(C.Structure):
    pass


RcData._fields_ = [("param", C.c_char_p), ("value", C.c_char_p), ("section_reqd", C.c_char_p), ("m", C.c_int), ("n", C.c_int),
                   ("next", C.POINTER(RcData))]
_P = C.POINTER(RcData)


def lib():
    """The reference-built parser, or None if oracle/_ref was never built."""
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_ref", "libreadconfig_ref.so")
        if not os.path.exists(so):
            return None
        r = C.CDLL(so)
        r.rc_read.restype = _P
        r.rc_read.argtypes = [C.c_char_p, C.c_void_p]
        r.rc_append.argtypes = [_P, C.c_char_p, C.c_void_p]
        r.rc_register_files.argtypes = [_P, C.c_int, C.POINTER(C.c_char_p)]
        r.rc_register_args.argtypes = [_P, C.c_int, C.POINTER(C.c_char_p)]
        r.rc_get_file.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        r.rc_sprint.restype = C.c_void_p
        r.rc_sprint.argtypes = [_P]
        r.rc_free.argtypes = [C.c_void_p]
        r.rc_clear.argtypes = [_P]
        r.rc_set_section.argtypes = [_P, C.c_char_p]
        r.rc_exists.argtypes = [_P, C.c_char_p]
        r.rc_get_boolean.argtypes = [_P, C.c_char_p]
        r.rc_assign_int.argtypes = [_P, C.c_char_p, C.POINTER(C.c_int)]
        r.rc_assign_real.argtypes = [_P, C.c_char_p, C.POINTER(C.c_double)]
        r.rc_get_string.restype = C.c_void_p
        r.rc_get_string.argtypes = [_P, C.c_char_p]
        r.rc_get_substring.restype = C.c_void_p
        r.rc_get_substring.argtypes = [_P, C.c_char_p, C.c_int]
        r.rc_size.argtypes = [_P, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        r.rc_get_real_vector.restype = C.POINTER(C.c_double)
        r.rc_get_real_vector.argtypes = [_P, C.c_char_p, C.POINTER(C.c_int)]
        r.rc_get_int_vector.restype = C.POINTER(C.c_int)
        r.rc_get_int_vector.argtypes = [_P, C.c_char_p, C.POINTER(C.c_int)]
        _LIB = r
    return _LIB


def _take(r, p):
    if not p:
        return None
    s = C.string_at(p).decode("utf-8", "surrogateescape")
    r.rc_free(p)
    return s


def split_scopes(param):
    """(scope, name) pairs a tool could use to reach `param`: unscoped and every section prefix."""
    out = [(None, param)]
    parts = param.split(".")
    for k in range(1, len(parts)):
        out.append((".".join(parts[:k]), ".".join(parts[k:])))
    return out


def lookups(entries, extra=()):
    keys = []
    for param, _, _, _ in entries:
        for sc in split_scopes(param):
            if sc not in keys:
                keys.append(sc)
            up = (sc[0], sc[1].upper())
            if up not in keys:
                keys.append(up)
    for sc in extra:
        if sc not in keys:
            keys.append(sc)
    return keys


def dump(cfg_path=None, argv=None, extra_lookups=()):
    """Parse with the reference and return {"entries", "sprint", "reads"}.  argv: DataFileEngineCfg.cpp:61-80."""
    r = lib()
    b = lambda s: None if s is None else s.encode()
    if argv is not None:
        arr = (C.c_char_p * len(argv))(*[b(a) for a in argv])
        data = r.rc_read(None, None)
        r.rc_register_files(data, len(argv), arr)
        ifile = r.rc_get_file(len(argv), arr)
        if ifile:
            if not r.rc_append(data, arr[ifile], None):
                return {"error": True}
        r.rc_register_args(data, len(argv), arr)
    else:
        data = r.rc_read(b(cfg_path), None)
        if not data:
            return {"error": True}
    entries = []
    node = data
    while node and node.contents.param is not None:
        d = node.contents
        entries.append([d.param.decode("utf-8", "surrogateescape"),
                        None if d.value is None else d.value.decode("utf-8", "surrogateescape"), d.m, d.n])
        node = d.next
    reads = []
    for scope, name in lookups(entries, extra_lookups):
        r.rc_set_section(data, b(scope))
        iv, dv, m, n, ln = C.c_int(-12345), C.c_double(-12345.5), C.c_int(), C.c_int(), C.c_int()
        rec = {"scope": scope, "name": name, "exists": int(r.rc_exists(data, b(name))), "boolean": int(r.rc_get_boolean(data, b(name)))}
        rec["int"] = iv.value if r.rc_assign_int(data, b(name), C.byref(iv)) else None
        rec["real"] = dv.value if r.rc_assign_real(data, b(name), C.byref(dv)) else None
        rec["string"] = _take(r, r.rc_get_string(data, b(name)))
        cnt = r.rc_size(data, b(name), C.byref(m), C.byref(n))
        rec["size"] = [cnt, m.value, n.value]
        subs, i = [], 0
        while True:
            s = _take(r, r.rc_get_substring(data, b(name), i))
            if s is None or i > 64:
                break
            subs.append(s)
            i += 1
        rec["substrings"] = subs
        pv = r.rc_get_real_vector(data, b(name), C.byref(ln))
        rec["real_vector"] = [pv[k] for k in range(ln.value)] if pv else None
        if pv:
            r.rc_free(pv)
        pi = r.rc_get_int_vector(data, b(name), C.byref(ln))
        rec["int_vector"] = [pi[k] for k in range(ln.value)] if pi else None
        if pi:
            r.rc_free(pi)
        reads.append(rec)
    r.rc_set_section(data, None)
    sp = _take(r, r.rc_sprint(data))
    r.rc_clear(data)
    return {"entries": entries, "sprint": sp, "reads": reads}
