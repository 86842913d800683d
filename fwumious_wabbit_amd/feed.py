"""Host-side mirror of the reference's input layer over the C ABI: VwNamespaceMap (vwmap.rs), VowpalParser (parser.rs)
and RecordCache (cache.rs).  Same names and error behaviour as the reference so that the tests read like its own."""
import ctypes as C

import numpy as np

from . import _capi as capi


class FlushCommand(Exception):
    """parser.rs:31-45: not an error, the client's "flush" message"""


class HogwildLoadCommand(Exception):
    """parser.rs:33-57"""

    def __init__(self, filename):
        super().__init__(f'Not really an error: a "hogwild_load" command from client to load: {filename}')
        self.filename = filename


class VwNamespaceMap:
    def __init__(self, data: str = None, _handle=None):  # vwmap.rs:106 VwNamespaceMap::new(csv text)
        self.L = capi.lib()
        self.h = C.c_void_p()
        if _handle is not None:
            self.h = _handle
        else:
            b = data.encode()
            capi.check(self.L.fwgpu_vwmap_from_csv(b, len(b), C.byref(self.h)))

    @classmethod
    def new_from_csv_filepath(cls, path):  # vwmap.rs:91-104
        with open(path, "r") as f:
            return cls(f.read())

    @classmethod
    def new_from_buf(cls, json_bytes: bytes):  # persistence.rs:44-52 (without the u64 length prefix)
        L = capi.lib()
        h = C.c_void_p()
        capi.check(L.fwgpu_vwmap_from_json(json_bytes, len(json_bytes), C.byref(h)))
        return cls(_handle=h)

    @property
    def num_namespaces(self):
        return int(self.L.fwgpu_vwmap_num_namespaces(self.h))

    def to_json(self) -> bytes:  # serde_json::to_vec_pretty(&vw_source)
        n = C.c_uint64()
        capi.check(self.L.fwgpu_vwmap_to_json(self.h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        capi.check(self.L.fwgpu_vwmap_to_json(self.h, buf, n.value, C.byref(n)))
        return buf.raw[: n.value]

    def lookup(self, name: str, verbose=False):
        """(namespace_index, is_f32) of a vw name / verbose name"""
        idx, f32 = C.c_uint32(), C.c_uint32()
        b = name.encode()
        capi.check(self.L.fwgpu_vwmap_lookup(self.h, b, len(b), int(verbose), C.byref(idx), C.byref(f32)))
        return idx.value, bool(f32.value)

    def close(self):
        if self.h:
            self.L.fwgpu_vwmap_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VowpalParser:
    def __init__(self, vw: VwNamespaceMap):  # parser.rs:78-105
        self.L = capi.lib()
        self.h = C.c_void_p()
        capi.check(self.L.fwgpu_parser_create(vw.h, C.byref(self.h)))
        self._buf = np.zeros(1 << 16, dtype=np.uint32)

    def _result(self, rc, n):
        if rc == capi.PARSE_FLUSH:
            raise FlushCommand()
        if rc == capi.PARSE_HOGWILD_LOAD:
            raise HogwildLoadCommand(self.L.fwgpu_parser_command_argument(self.h).decode())
        capi.check(rc)
        return self._buf[: n.value].copy()

    def next_vowpal(self, line: bytes) -> np.ndarray:
        """parser.rs:166-176 on one line (bytes up to and including the newline); b"" = end of stream -> empty record"""
        n = C.c_uint32()
        rc = self.L.fwgpu_parser_parse_line(self.h, line, len(line), capi.ptr(self._buf), self._buf.size, C.byref(n))
        return self._result(rc, n)

    def next_vowpal_with_cache(self, cached: bytes, line: bytes) -> np.ndarray:  # parser.rs:195-211
        n = C.c_uint32()
        rc = self.L.fwgpu_parser_parse_with_prefix(self.h, cached, len(cached), line, len(line), capi.ptr(self._buf),
                                                   self._buf.size, C.byref(n))
        return self._result(rc, n)

    def scan_context(self, cached: bytes) -> "ParsePrefix":
        """The context line scanned once; `next_vowpal_after` then equals next_vowpal_with_cache(cached, line)."""
        return ParsePrefix(self, cached)

    def next_vowpal_after(self, prefix: "ParsePrefix", line: bytes) -> np.ndarray:
        n = C.c_uint32()
        rc = self.L.fwgpu_parser_parse_after_prefix(self.h, prefix.h, line, len(line), capi.ptr(self._buf), self._buf.size,
                                                    C.byref(n))
        return self._result(rc, n)

    def next_vowpal_candidate(self, prefix: "ParsePrefix", line: bytes):
        """(record, is_delta): the candidate-only form of next_vowpal_after's record when the request can be split off the
        context's record (fwgpu_parser_parse_candidate), else the merged record"""
        n, d = C.c_uint32(), C.c_int32()
        rc = self.L.fwgpu_parser_parse_candidate(self.h, prefix.h, line, len(line), capi.ptr(self._buf), self._buf.size,
                                                 C.byref(n), C.byref(d))
        return self._result(rc, n), bool(d.value)

    def parse_buffer(self, text: bytes, max_records=1 << 20, words_cap=None):
        """Many lines -> (records u32[], rec_off u64[n+1], bytes consumed, status).  Stops at the first line that is
        not an example; status is OK / PARSE_FLUSH / PARSE_HOGWILD_LOAD / ERR_PARSE."""
        words_cap = words_cap or max(1024, len(text))  # a record never has more words than its line has bytes + header
        words = np.empty(words_cap + 64 * 1024, dtype=np.uint32)
        off = np.zeros(max_records + 1, dtype=np.uint64)
        nr, nw, used = C.c_uint64(), C.c_uint64(), C.c_uint64()
        rc = self.L.fwgpu_parser_parse_buffer(self.h, text, len(text), capi.ptr(words), words.size, capi.ptr(off),
                                              max_records, C.byref(nr), C.byref(nw), C.byref(used))
        return words[: nw.value].copy(), off[: nr.value + 1].copy(), used.value, rc

    def close(self):
        if self.h:
            self.L.fwgpu_parser_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ParsePrefix:
    def __init__(self, parser: VowpalParser, cached: bytes):
        self.L = capi.lib()
        self.h = C.c_void_p()
        capi.check(self.L.fwgpu_parse_prefix_create(parser.h, cached, len(cached), C.byref(self.h)))

    @property
    def resumable(self) -> bool:
        return bool(self.L.fwgpu_parse_prefix_resumable(self.h))

    def is_record(self, record: np.ndarray) -> bool:
        """the scanned part of the context is all of it: `record` (the context line parsed on its own) is the state requests resume from"""
        r = np.ascontiguousarray(record, dtype=np.uint32)
        return bool(self.L.fwgpu_parse_prefix_is_record(self.h, capi.ptr(r), len(r)))

    def close(self):
        if self.h:
            self.L.fwgpu_parse_prefix_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RecordCache:
    def __init__(self, input_filename: str, enabled: bool, vw_map: VwNamespaceMap):  # cache.rs:70-131
        self.L = capi.lib()
        self.h = C.c_void_p()
        self.reading = self.writing = False
        if enabled:
            capi.check(self.L.fwgpu_cache_open(input_filename.encode(), vw_map.h, C.byref(self.h)))
            self.reading = bool(self.L.fwgpu_cache_is_reading(self.h))
            self.writing = bool(self.L.fwgpu_cache_is_writing(self.h))

    def push_record(self, record_buf: np.ndarray):  # cache.rs:133-144
        if self.writing:
            r = np.ascontiguousarray(record_buf, dtype=np.uint32)
            capi.check(self.L.fwgpu_cache_push_records(self.h, capi.ptr(r), r.size))

    push_records = push_record  # any number of whole records, back to back

    def write_finish(self):  # cache.rs:146-152
        if self.h:
            capi.check(self.L.fwgpu_cache_write_finish(self.h))
            self.writing = False

    def next_records(self, words_cap=1 << 22, max_records=1 << 16):
        """bulk get_next_record (cache.rs:187-232): (records, rec_off); empty at end of file"""
        if not self.reading:
            raise capi.FwgpuError(capi.ERR_INVALID if hasattr(capi, "ERR_INVALID") else 1,
                                  "next_recrod() called on reading cache, when not opened in reading mode")
        words = np.empty(words_cap, dtype=np.uint32)
        off = np.zeros(max_records + 1, dtype=np.uint64)
        nr, nw = C.c_uint64(), C.c_uint64()
        capi.check(self.L.fwgpu_cache_next_records(self.h, capi.ptr(words), words.size, capi.ptr(off), max_records,
                                                   C.byref(nr), C.byref(nw)))
        return words[: nw.value].copy(), off[: nr.value + 1].copy()

    def close(self):
        if self.h:
            self.L.fwgpu_cache_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_buffered_input(input_filename: str):
    """buffer_handler.rs:8-37: a reader over .vw / .gz / .zst input; iterate for chunks of decompressed bytes"""
    L = capi.lib()
    h = C.c_void_p()
    capi.check(L.fwgpu_input_open(input_filename.encode(), C.byref(h)))

    def chunks(size=1 << 20):
        buf = C.create_string_buffer(size)
        n = C.c_uint64()
        try:
            while True:
                capi.check(L.fwgpu_input_read(h, buf, size, C.byref(n)))
                if n.value == 0:
                    return
                yield buf.raw[: n.value]
        finally:
            L.fwgpu_input_close(h)

    return chunks()
