"""Configuration of the tools: `exe [key=value ...] [file.cfg]`.

Host-side mirror of the reference's `DataFile config(argc, argv)` (src/include/DataFile.h:49-337 over
src/tools/DataFileEngineCfg.cpp and src/tools/readconfig.c) on top of the C ABI `ecckd_cfg_*`
(include/ecckd_hip.h, ecckd_amd/csrc/config.cpp).  The method names and the "value left untouched, returns
False when absent" contract follow DataFile::read; because Python has no reference arguments the value is
returned instead and `default` plays the role of the untouched variable.
"""
import ctypes as C

from . import _lib
from ._lib import check


def _b(s):
    return None if s is None else str(s).encode()


class Config:
    def __init__(self, argv=None, text=None, path=None):
        """argv: the full argument vector including argv[0] (DataFileEngineCfg.cpp:61-80); text / path: parse a
        configuration given directly."""
        self.lib = _lib.load_library()
        h = C.c_void_p()
        if argv is not None:
            arr = (C.c_char_p * len(argv))(*[_b(a) for a in argv])
            check(self.lib.ecckd_cfg_from_args(len(argv), arr, C.byref(h)))
        else:
            check(self.lib.ecckd_cfg_create(C.byref(h)))
        self.handle = h
        if path is not None:
            check(self.lib.ecckd_cfg_append_file(h, _b(path)))
        if text is not None:
            check(self.lib.ecckd_cfg_append_text(h, _b(text), b"<text>"))

    def close(self):
        if self.handle:
            self.lib.ecckd_cfg_destroy(self.handle)
            self.handle = None

    __del__ = close

    # ---- raw view ----
    def _string_call(self, fn, *args):
        n = C.c_size_t()
        check(fn(self.handle, *args, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value + 1)
        check(fn(self.handle, *args, buf, n.value + 1, C.byref(n)))
        return buf.raw[:n.value].decode("utf-8", "surrogateescape")

    @property
    def file_name(self):
        return self._string_call(self.lib.ecckd_cfg_file_name)

    def entries(self):
        """[(param, value or None, m, n)] in the order of definition."""
        n = C.c_int()
        check(self.lib.ecckd_cfg_count(self.handle, C.byref(n)))
        out = []
        for i in range(n.value):
            vl, hv, m, k = C.c_size_t(), C.c_int(), C.c_int(), C.c_int()
            pbuf = C.create_string_buffer(4096)
            check(self.lib.ecckd_cfg_entry(self.handle, i, pbuf, 4096, None, 0, C.byref(vl), C.byref(hv), C.byref(m), C.byref(k)))
            vbuf = C.create_string_buffer(vl.value + 1)
            check(self.lib.ecckd_cfg_entry(self.handle, i, pbuf, 4096, vbuf, vl.value + 1, C.byref(vl), C.byref(hv), C.byref(m),
                                           C.byref(k)))
            out.append((pbuf.value.decode("utf-8", "surrogateescape"),
                        vbuf.raw[:vl.value].decode("utf-8", "surrogateescape") if hv.value else None, m.value, k.value))
        return out

    def sprint(self):
        """rc_sprint: what the tools store in the `config` attribute of their output files."""
        return self._string_call(self.lib.ecckd_cfg_sprint)

    def register(self, param, value=None):
        check(self.lib.ecckd_cfg_register(self.handle, _b(param), _b(value)))

    # ---- DataFile::exist / read ----
    def exist(self, name, scope=None):
        e = C.c_int()
        check(self.lib.ecckd_cfg_exists(self.handle, _b(scope), _b(name), C.byref(e)))
        return bool(e.value)

    def read_bool(self, name, scope=None):
        v = C.c_int()
        check(self.lib.ecckd_cfg_get_boolean(self.handle, _b(scope), _b(name), C.byref(v)))
        return bool(v.value)

    def read_int(self, name, scope=None, default=None):
        v, f = C.c_int(), C.c_int()
        check(self.lib.ecckd_cfg_get_int(self.handle, _b(scope), _b(name), C.byref(v), C.byref(f)))
        return v.value if f.value else default

    def read_real(self, name, scope=None, default=None):
        v, f = C.c_double(), C.c_int()
        check(self.lib.ecckd_cfg_get_real(self.handle, _b(scope), _b(name), C.byref(v), C.byref(f)))
        return v.value if f.value else default

    def read_string(self, name, scope=None, index=-1, default=None):
        n, f = C.c_size_t(), C.c_int()
        check(self.lib.ecckd_cfg_get_string(self.handle, _b(scope), _b(name), index, None, 0, C.byref(n), C.byref(f)))
        if not f.value:
            return default
        buf = C.create_string_buffer(n.value + 1)
        check(self.lib.ecckd_cfg_get_string(self.handle, _b(scope), _b(name), index, buf, n.value + 1, C.byref(n), C.byref(f)))
        return buf.raw[:n.value].decode("utf-8", "surrogateescape")

    def size(self, name, scope=None):
        """(number of items, declared m, declared n) - rc_size."""
        c, m, n = C.c_int(), C.c_int(), C.c_int()
        check(self.lib.ecckd_cfg_size(self.handle, _b(scope), _b(name), C.byref(c), C.byref(m), C.byref(n)))
        return c.value, m.value, n.value

    def read_strings(self, name, scope=None):
        """All items of a list-valued parameter (the tools loop DataFile::read(str, name, i) until it fails)."""
        out, i = [], 0
        while True:
            s = self.read_string(name, scope, i)
            if s is None:
                return out
            out.append(s)
            i += 1

    def read_real_vector(self, name, scope=None, default=None):
        n = C.c_int()
        check(self.lib.ecckd_cfg_get_real_vector(self.handle, _b(scope), _b(name), None, 0, C.byref(n)))
        if n.value == 0:
            return default
        buf = (C.c_double * n.value)()
        check(self.lib.ecckd_cfg_get_real_vector(self.handle, _b(scope), _b(name), buf, n.value, C.byref(n)))
        return list(buf)

    def read_int_vector(self, name, scope=None, default=None):
        n = C.c_int()
        check(self.lib.ecckd_cfg_get_int_vector(self.handle, _b(scope), _b(name), None, 0, C.byref(n)))
        if n.value == 0:
            return default
        buf = (C.c_int * n.value)()
        check(self.lib.ecckd_cfg_get_int_vector(self.handle, _b(scope), _b(name), buf, n.value, C.byref(n)))
        return list(buf)
