/*
 * qzo.h — CPU oracle for the QATzip software-fallback hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / the timed CPU baseline.
 *
 * What it restates (reference = intel/QATzip @ /root/reference):
 *   - qzDeflateSWCompress   src/qatzip_sw.c:77-256   (one zlib stream per call,
 *     Z_FULL_FLUSH per hw_buff_sz chunk, Z_FINISH on the last, header patching)
 *   - qzDeflateSWDecompress / qzSWDecompressMultiGzip   src/qatzip_sw.c:258-441
 *   - qzLZ4SWCompress / qzLZ4SWDecompress               src/qatzip_sw.c:443-577
 * The arithmetic those call sites delegate to lives in third-party libraries
 * that are NOT vendored in the reference tree:
 *   zlib  (configure.ac:98-107 floor 1.2.7; pinned here to 1.2.11)
 *         deflateInit2(level, Z_DEFLATED, wbits, memLevel 9, Z_DEFAULT_STRATEGY)
 *   lz4   (README.md:117-119 floor 1.8.3; pinned here to 1.9.3)
 *         LZ4F_compressFrame / LZ4F_decompress
 * Their published algorithms (RFC 1950/1951/1952, the LZ4 block + frame
 * formats, zlib's deflate_fast/trees heuristics, lz4's LZ4_compress_fast) are
 * restated here from scratch in plain C.
 *
 * Pinning: tests/golden/ holds vectors produced in the build container by the
 * system libz 1.2.11 / liblz4 1.9.3 driven exactly as the reference call sites
 * drive them (tests/golden/gen_golden.py); tests/test_oracle.py checks every
 * function here against them, plus the reference's own CRC known-answer
 * (test/main.c:4283-4337: crc out-param == zlib crc32(src)).
 */
#ifndef QZO_H
#define QZO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* data formats, numbering of DataFormatInternal_T (src/qatzip_internal.h:238-253) */
enum { QZO_DEFLATE_4B = 0, QZO_DEFLATE_GZIP, QZO_DEFLATE_GZIP_EXT, QZO_DEFLATE_RAW,
       QZO_LZ4_FH, QZO_LZ4S_BK, QZO_DEFLATE_ZLIB };

/* checksums */
uint32_t qzo_crc32(uint32_t crc, const uint8_t *p, size_t n);         /* zlib crc32() */
uint32_t qzo_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);
uint32_t qzo_adler32(uint32_t adler, const uint8_t *p, size_t n);
uint32_t qzo_xxh32(const uint8_t *p, size_t n, uint32_t seed);

/* raw deflate of ONE chunk with fresh state, zlib levels 1..9 (deflate_fast 1-3, deflate_slow 4-9),
 * memLevel 9, wbits 15.  final=0 ends with the Z_FULL_FLUSH marker, final=1 with
 * BFINAL.  Returns bytes written or (size_t)-1 on overflow. */
size_t qzo_deflate_chunk(const uint8_t *src, size_t n, uint8_t *dst, size_t cap,
                         int level, int final);

/* LZ77 symbol dump of one chunk (for kernel-level parity): lc[i], dist[i]
 * (dist==0 => literal lc; else length = lc+3).  Returns symbol count. */
size_t qzo_deflate_symbols(const uint8_t *src, size_t n, int level,
                           uint8_t *lc, uint16_t *dist, size_t cap);

/* qzDeflateSWCompress restated: one call = one stream. `last` as in qzCompress.
 * crc: in/out like qz_sess->crc32 (may be NULL).  Returns 0 (QZ_OK) / -2 (QZ_FAIL). */
int qzo_sw_compress(int fmt, int level, uint32_t hw_buff_sz,
                    const uint8_t *src, uint32_t *src_len,
                    uint8_t *dst, uint32_t *dst_len, int last, unsigned long *crc);

/* raw inflate: returns 0 ok(stream end), 1 = need more output/input (stopped), <0 data error.
 * *in_used / *out_used report progress. stop_at_sync unused. */
int qzo_inflate_raw(const uint8_t *src, size_t n, uint8_t *dst, size_t cap,
                    size_t *in_used, size_t *out_used);

/* qzSWDecompressMulti restated for complete members/frames in src.
 * Returns QZ_OK(0), QZ_DATA_ERROR(-4), QZ_FAIL(-2). */
int qzo_sw_decompress(int fmt, const uint8_t *src, uint32_t *src_len,
                      uint8_t *dst, uint32_t *dst_len);

/* LZ4 block (LZ4_compress_default semantics, dstCapacity limited) — returns
 * compressed size or 0 when it does not fit (=> stored block in the frame). */
int qzo_lz4_compress_block(const uint8_t *src, int n, uint8_t *dst, int cap);
int qzo_lz4_decompress_block(const uint8_t *src, int n, uint8_t *dst, int cap);
int qzo_lz4_decompress_block_prefix(const uint8_t *src, int n, uint8_t *dst, int cap, size_t prefix);
/* LZ4F_compressFrame with {contentChecksum=1, contentSize=n, autoFlush=1, level<3}.
 * Returns frame size, 0 on error. */
size_t qzo_lz4f_compress_frame(const uint8_t *src, size_t n, uint8_t *dst, size_t cap);
size_t qzo_lz4f_bound(size_t n);

#ifdef __cplusplus
}
#endif
#endif
