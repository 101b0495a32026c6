"""Host logic of the qatzip.h surface that needs no GPU: defaults, parameter validation
(the cases of the reference's param test, test/main.c:1162-1251), qzMaxCompressedLength
(SURVEY §8b table), pinned-memory bookkeeping and the exported symbol list."""
import ctypes as C
import os
import re

from qatzip_amd import api as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_api_symbol_is_exported():
    L = A.lib()
    txt = open(os.path.join(ROOT, "include", "qatzip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = sorted(set(re.findall(r"\b(qz[A-Z]\w*)\s*\(", txt)))
    assert "qzCompress" in names and "qzDecompressStream" in names and len(names) > 55
    for n in names:
        assert hasattr(L, n), "missing export: " + n


def test_defaults_match_reference():
    L = A.lib()
    p = A.QzSessionParams()
    assert L.qzGetDefaults(C.byref(p)) == A.QZ_OK
    assert (p.huffman_hdr, p.direction, p.data_fmt, p.comp_lvl, p.comp_algorithm) == (0, 2, 2, 1, 8)
    assert (p.sw_backup, p.hw_buff_sz, p.strm_buff_sz, p.input_sz_thrshold) == (1, 65536, 65536, 1024)
    assert (p.max_forks, p.req_cnt_thrshold, p.wait_cnt_thrshold) == (3, 32, 8)
    assert L.qzGetDefaults(None) == A.QZ_PARAMS


def test_set_defaults_rejects_what_the_reference_rejects():
    L = A.lib()

    def bad(**kw):
        p = A.QzSessionParams(); L.qzGetDefaults(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        return L.qzSetDefaults(C.byref(p))

    assert bad(huffman_hdr=2) == A.QZ_PARAMS            # test/main.c:1162
    assert bad(direction=3) == A.QZ_PARAMS
    assert bad(comp_lvl=0) == A.QZ_PARAMS
    assert bad(comp_lvl=10) == A.QZ_PARAMS
    assert bad(sw_backup=2) == A.QZ_PARAMS
    assert bad(hw_buff_sz=0) == A.QZ_PARAMS
    assert bad(hw_buff_sz=1025) == A.QZ_PARAMS          # not a power of two
    assert bad(hw_buff_sz=2 * 1024 * 1024) == A.QZ_PARAMS
    assert bad(strm_buff_sz=100) == A.QZ_PARAMS
    assert bad(input_sz_thrshold=100) == A.QZ_PARAMS
    assert bad(comp_algorithm=ord("4")) == A.QZ_PARAMS
    assert bad() == A.QZ_OK


def test_setup_session_without_gpu_and_duplicates():
    L = A.lib()
    s = A.QzSession()
    assert L.qzSetupSession(None, None) == A.QZ_PARAMS
    assert L.qzSetupSession(C.byref(s), None) == A.QZ_OK
    assert s.internal
    assert L.qzSetupSession(C.byref(s), None) == A.QZ_DUPLICATE       # src/qatzip.c:1143-1145
    assert L.qzTeardownSession(C.byref(s)) == A.QZ_OK and not s.internal
    assert L.qzTeardownSession(None) == A.QZ_PARAMS
    assert L.qzInit(None, 1) == A.QZ_PARAMS and L.qzInit(C.byref(s), 2) == A.QZ_PARAMS


def test_arg_checks_on_hot_path_entry_points():
    L = A.lib()
    s = A.QzSession()
    sl, dl = C.c_uint(5), C.c_uint(100)
    dst = C.create_string_buffer(100)
    assert L.qzCompress(C.byref(s), b"hello", C.byref(sl), dst, C.byref(dl), 2) == A.QZ_PARAMS   # last not in {0,1}
    assert sl.value == 0 and dl.value == 0                                                      # both zeroed
    sl, dl = C.c_uint(5), C.c_uint(100)
    assert L.qzCompress(None, b"hello", C.byref(sl), dst, C.byref(dl), 1) == A.QZ_PARAMS
    sl, dl = C.c_uint(0), C.c_uint(100)
    assert L.qzDecompress(C.byref(s), b"", C.byref(sl), dst, C.byref(dl)) == A.QZ_OK and dl.value == 0  # :2465-2468


def test_max_compressed_length_values():
    L = A.lib()
    assert L.qzMaxCompressedLength(0, None) == 34
    assert L.qzMaxCompressedLength(65536, None) == 73808
    assert L.qzMaxCompressedLength(200000, None) == 225080
    assert L.qzMaxCompressedLength(0xffffffff, None) == 0          # overflow => 0


def test_qzmalloc_common_falls_back_to_malloc_without_gpu():
    L = A.lib()
    p = L.qzMalloc(4096, -1, A.COMMON_MEM)
    assert p
    C.memset(p, 7, 4096)
    L.qzFree(p)
    L.qzFree(None)                                                  # no-op, src/qatzip_mem.c:228-230
    assert L.qzMemFindAddr(12345) == 0
