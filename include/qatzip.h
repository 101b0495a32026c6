/*
 * qatzip.h — the QATzip application interface, as exported by libqatzip_amd.so.
 *
 * This header is written for the MI355X backend; it declares the same types,
 * constants and entry points (same names, layouts, argument meaning and return
 * codes) as intel/QATzip's include/qatzip.h API 2.5, so that a program built
 * against the original header links and runs against this library unchanged.
 * Each group below cites the reference declaration it stands in for
 * (file:line in /root/reference).  The behaviour behind the hot-path entry
 * points is documented in DESIGN.md / INTEGRATION.md; everything that only
 * makes sense for Intel QAT silicon (metadata blobs, CRC64 configs, LZ4s) is
 * declared for link compatibility and returns QZ_NOT_SUPPORTED.
 */
#ifndef _QATZIP_H
#define _QATZIP_H
#ifdef __cplusplus
extern "C" {
#endif
#include <stdint.h>
#include <string.h>

#define QATZIP_API_VERSION_NUM_MAJOR (2)
#define QATZIP_API_VERSION_NUM_MINOR (5)
#define QATZIP_API_VERSION (QATZIP_API_VERSION_NUM_MAJOR * 10000 + QATZIP_API_VERSION_NUM_MINOR * 100)
#define QATZIP_API

/* ---- enumerations: include/qatzip.h:166-300 ---- */
typedef enum QzHuffmanHdr_E { QZ_DYNAMIC_HDR = 0, QZ_STATIC_HDR } QzHuffmanHdr_T;
typedef enum PinMem_E { COMMON_MEM = 0, PINNED_MEM } PinMem_T;
typedef enum QzDirection_E { QZ_DIR_COMPRESS = 0, QZ_DIR_DECOMPRESS, QZ_DIR_BOTH } QzDirection_T;
typedef enum QzDataFormat_E {
    QZ_DEFLATE_4B = 0, QZ_DEFLATE_GZIP, QZ_DEFLATE_GZIP_EXT, QZ_DEFLATE_RAW, QZ_FMT_NUM
} QzDataFormat_T;
typedef enum QzPollingMode_E { QZ_PERIODICAL_POLLING = 0, QZ_BUSY_POLLING } QzPollingMode_T;
typedef enum QzCrcType_E { QZ_CRC32 = 0, QZ_ADLER, NONE } QzCrcType_T;
typedef enum QzSoftwareComponentType_E {
    QZ_COMPONENT_FIRMWARE = 0, QZ_COMPONENT_KERNEL_DRIVER, QZ_COMPONENT_USER_DRIVER,
    QZ_COMPONENT_QATZIP_API, QZ_COMPONENT_SOFTWARE_PROVIDER
} QzSoftwareComponentType_T;

/* ---- return codes: include/qatzip.h:311-361 (0 ok, <0 error, >0 informational) ---- */
#define QZ_OK (0)
#define QZ_DUPLICATE (1)
#define QZ_FORCE_SW (2)
#define QZ_PARAMS (-1)
#define QZ_FAIL (-2)
#define QZ_BUF_ERROR (-3)
#define QZ_DATA_ERROR (-4)
#define QZ_TIMEOUT (-5)
#define QZ_INTEG (-100)
#define QZ_NO_HW (11)
#define QZ_NO_MDRV (12)
#define QZ_NO_INST_ATTACH (13)
#define QZ_LOW_MEM (14)
#define QZ_LOW_DEST_MEM (15)
#define QZ_UNSUPPORTED_FMT (16)
#define QZ_NONE (100)
#define QZ_NOSW_NO_HW (-101)
#define QZ_NOSW_NO_MDRV (-102)
#define QZ_NOSW_NO_INST_ATTACH (-103)
#define QZ_NOSW_LOW_MEM (-104)
#define QZ_NO_SW_AVAIL (-105)
#define QZ_NOSW_UNSUPPORTED_FMT (-116)
#define QZ_POST_PROCESS_ERROR (-117)
#define QZ_METADATA_OVERFLOW (-118)
#define QZ_OUT_OF_RANGE (-119)
#define QZ_NOT_SUPPORTED (-200)

#define QZ_MAX_ALGORITHMS ((int)255)
#define QZ_DEFLATE ((unsigned char)8)
#define QZ_LZ4 ((unsigned char)'4')
#define QZ_LZ4_BLOCK ((unsigned char)'B')
#define QZ_LZ4s ((unsigned char)'s')
#define QZ_ZSTD ((unsigned char)'Z')

#ifndef MIN
#define MIN(a, b) (((a) < (b)) ? (a) : (b))
#endif
#define QZ_MEMCPY(dest, src, dest_sz, src_sz) memcpy((void *)(dest), (void *)(src), (size_t)MIN(dest_sz, src_sz))

typedef int (*qzLZ4SCallbackFn)(void *external, const unsigned char *src, unsigned int *src_len,
                                unsigned char *dest, unsigned int *dest_len, int *ExtStatus);

/* ---- session parameters: include/qatzip.h:461-571 (field order is ABI) ---- */
typedef struct QzSessionParams_S {
    QzHuffmanHdr_T huffman_hdr;
    QzDirection_T direction;
    QzDataFormat_T data_fmt;
    unsigned int comp_lvl;
    unsigned char comp_algorithm;
    unsigned int max_forks;
    unsigned char sw_backup;
    unsigned int hw_buff_sz;
    unsigned int strm_buff_sz;
    unsigned int input_sz_thrshold;
    unsigned int req_cnt_thrshold;
    unsigned int wait_cnt_thrshold;
} QzSessionParams_T;

typedef struct QzSessionParamsCommon_S {
    QzDirection_T direction;
    unsigned int comp_lvl;
    unsigned char comp_algorithm;
    unsigned int max_forks;
    unsigned char sw_backup;
    unsigned int hw_buff_sz;
    unsigned int strm_buff_sz;
    unsigned int input_sz_thrshold;
    unsigned int req_cnt_thrshold;
    unsigned int wait_cnt_thrshold;
    QzPollingMode_T polling_mode;
    unsigned int is_sensitive_mode;
} QzSessionParamsCommon_T;

typedef struct QzSessionParamsDeflate_S {
    QzSessionParamsCommon_T common_params;
    QzHuffmanHdr_T huffman_hdr;
    QzDataFormat_T data_fmt;
} QzSessionParamsDeflate_T;

typedef struct QzSessionParamsLZ4_S { QzSessionParamsCommon_T common_params; } QzSessionParamsLZ4_T;

typedef struct QzSessionParamsLZ4S_S {
    QzSessionParamsCommon_T common_params;
    qzLZ4SCallbackFn qzCallback;
    void *qzCallback_external;
    unsigned int lz4s_mini_match;
} QzSessionParamsLZ4S_T;

typedef struct QzSessionParamsDeflateExt_S {
    QzSessionParamsDeflate_T deflate_params;
    unsigned char stop_decompression_stream_end;
    unsigned char zlib_format;
} QzSessionParamsDeflateExt_T;

/* ---- defaults and limits: include/qatzip.h:573-600 ---- */
#define QZ_HUFF_HDR_DEFAULT QZ_DYNAMIC_HDR
#define QZ_DIRECTION_DEFAULT QZ_DIR_BOTH
#define QZ_DATA_FORMAT_DEFAULT QZ_DEFLATE_GZIP_EXT
#define QZ_COMP_LEVEL_DEFAULT 1
#define QZ_COMP_ALGOL_DEFAULT QZ_DEFLATE
#define QZ_POLL_SLEEP_DEFAULT 10
#define QZ_MAX_FORK_DEFAULT 3
#define QZ_SW_BACKUP_DEFAULT 1
#define QZ_HW_BUFF_SZ (64 * 1024)
#define QZ_HW_BUFF_SZ_Gen3 (1 * 1024 * 1024)
#define QZ_HW_BUFF_MIN_SZ (1 * 1024)
#define QZ_HW_BUFF_MAX_SZ (512 * 1024)
#define QZ_HW_BUFF_MAX_SZ_Gen3 (2 * 1024 * 1024 * 1024U)
#define QZ_STRM_BUFF_SZ_DEFAULT QZ_HW_BUFF_SZ
#define QZ_STRM_BUFF_MIN_SZ (1 * 1024)
#define QZ_STRM_BUFF_MAX_SZ (2 * 1024 * 1024 - 5 * 1024)
#define QZ_COMP_THRESHOLD_DEFAULT 1024
#define QZ_COMP_THRESHOLD_MINIMUM 128
#define QZ_REQ_THRESHOLD_MINIMUM 1
#define QZ_REQ_THRESHOLD_MAXIMUM 32
#define QZ_REQ_THRESHOLD_DEFAULT QZ_REQ_THRESHOLD_MAXIMUM
#define QZ_WAIT_CNT_THRESHOLD_DEFAULT 8
#define QZ_DEFLATE_COMP_LVL_MINIMUM (1)
#define QZ_DEFLATE_COMP_LVL_MAXIMUM (9)
#define QZ_DEFLATE_COMP_LVL_MAXIMUM_Gen3 (12)
#define QZ_LZS_COMP_LVL_MINIMUM (1)
#define QZ_LZS_COMP_LVL_MAXIMUM (12)
#define QZ_AUTO_SELECT_NUMA_NODE (-1)

/* ---- ext_rc bits: include/qatzip.h:640-664 ---- */
#define QZ_SW_BACKUP_BIT_POSITION (0)
#define QZ_SW_FORCESW_BIT_POSITION (1)
#define QZ_ENABLE_SOFTWARE_BACKUP(v) ((v) |= (1 << QZ_SW_BACKUP_BIT_POSITION))
#define QZ_ENABLE_SOFTWARE_ONLY_EXECUTION(v) ((v) |= (1 << QZ_SW_FORCESW_BIT_POSITION))
#define QZ_DISABLE_SOFTWARE_BACKUP(v) ((v) &= ~(1 << QZ_SW_BACKUP_BIT_POSITION))
#define QZ_DISABLE_SOFTWARE_ONLY_EXECUTION(v) ((v) &= ~(1 << QZ_SW_FORCESW_BIT_POSITION))
#define QZ_SW_EXECUTION_BIT (4)
#define QZ_SW_EXECUTION_MASK (1 << QZ_SW_EXECUTION_BIT)
#define QZ_SW_EXECUTION(ret, ext_rc) (!(ret) && ((ext_rc) & QZ_SW_EXECUTION_MASK))
#define QZ_TIMEOUT_BIT (8)
#define QZ_TIMEOUT_MASK (1 << QZ_TIMEOUT_BIT)
#define QZ_HW_TIMEOUT(ret, ext_rc) (!(ret) && ((ext_rc) & QZ_TIMEOUT_MASK))
#define QZ_POST_PROCESS_FAIL_BIT (10)
#define QZ_POST_PROCESS_FAIL_MASK (1 << QZ_POST_PROCESS_FAIL_BIT)
#define QZ_POST_PROCESS_FAIL(ret, ext_rc) ((ret) && ((ext_rc) & QZ_POST_PROCESS_FAIL_MASK))

/* ---- session / status / results: include/qatzip.h:676-760, 2358-2379 ---- */
typedef struct QzSession_S {
    signed long int hw_session_stat;   /* QZ_OK once a GPU backs the session */
    int thd_sess_stat;                 /* result of the last request */
    void *internal;                    /* library owned */
    unsigned long total_in;
    unsigned long total_out;
} QzSession_T;

typedef struct QzStatus_S {
    unsigned short int qat_hw_count;   /* here: number of visible MI355X devices */
    unsigned char qat_service_init;
    unsigned char qat_mem_drvr;
    unsigned char qat_instance_attach;
    unsigned long int memory_alloced;
    unsigned char using_huge_pages;
    signed long int hw_session_status;
    unsigned char algo_sw[QZ_MAX_ALGORITHMS];
    unsigned char algo_hw[QZ_MAX_ALGORITHMS];
} QzStatus_T;

#define QZ_MAX_STRING_LENGTH 64
typedef struct QzSoftwareVersionInfo_S {
    QzSoftwareComponentType_T component_type;
    unsigned char component_name[QZ_MAX_STRING_LENGTH];
    unsigned int major_version, minor_version, patch_version, build_number;
    unsigned char reserved[52];
} QzSoftwareVersionInfo_T;

typedef struct QzCrc64Config_S {
    uint64_t polynomial, initial_value; uint32_t reflect_in, reflect_out; uint64_t xor_out;
} QzCrc64Config_T;
typedef struct QzCrc32Config_S {
    uint32_t polynomial, initial_value, reflect_in, reflect_out, xor_out;
} QzCrc32Config_T;

#define QZ_INPUT_CRC_VALID_BIT (4)
#define QZ_INPUT_CRC_VALID_MASK (1 << QZ_INPUT_CRC_VALID_BIT)
#define QZ_OUTPUT_CRC_VALID_BIT (8)
#define QZ_OUTPUT_CRC_VALID_MASK (1 << QZ_OUTPUT_CRC_VALID_BIT)
#define QZ_CRC32_VALID_BIT (12)
#define QZ_CRC32_VALID_MASK (1 << QZ_CRC32_VALID_BIT)
#define QZ_CRC64_VALID_BIT (16)
#define QZ_CRC64_VALID_MASK (1 << QZ_CRC64_VALID_BIT)

typedef struct QzCrcResult_S {
    int status; uint32_t valid_flags;
    union { uint32_t *crc_32; uint64_t *crc_64; } in_crc;
    union { uint32_t *crc_32; uint64_t *crc_64; } out_crc;
} QzCrcResult_T;

typedef struct QzResult_S {
    int status; void *cb_tag; unsigned int src_len, dest_len; uint64_t ext_rc;
    QzCrcResult_T *crc; void *extension_result;
} QzResult_T;
typedef int (*qzAsyncCallbackFn)(QzResult_T *res);
typedef void *QzMetadataBlob_T;

typedef enum QzLogLevel_E {
    LOG_NONE = 0, LOG_FATAL, LOG_ERROR, LOG_WARNING, LOG_INFO, LOG_DEBUG1, LOG_DEBUG2, LOG_DEBUG3
} QzLogLevel_T;

/* ---- life cycle: qzInit src/qatzip.c:630, qzSetupSession* :1118-1345, teardown :2673-2755 ---- */
QzLogLevel_T qzSetLogLevel(QzLogLevel_T level);
int qzInit(QzSession_T *sess, unsigned char sw_backup);
int qzSetupSession(QzSession_T *sess, QzSessionParams_T *params);
int qzSetupSessionDeflate(QzSession_T *sess, QzSessionParamsDeflate_T *params);
int qzSetupSessionLZ4(QzSession_T *sess, QzSessionParamsLZ4_T *params);
int qzSetupSessionLZ4S(QzSession_T *sess, QzSessionParamsLZ4S_T *params);
int qzSetupSessionDeflateExt(QzSession_T *sess, QzSessionParamsDeflateExt_T *params);
int qzTeardownSession(QzSession_T *sess);
int qzClose(QzSession_T *sess);
int qzGetStatus(QzSession_T *sess, QzStatus_T *status);
int qzGetDeflateEndOfStream(QzSession_T *sess, unsigned char *endofstream);

/* ---- the hot path: qzCompress* src/qatzip.c:1842-2097, qzDecompress* :2422-2671 ----
 * in:  *src_len = bytes available, *dest_len = capacity;  out: consumed / produced.
 * `last` must be 0 or 1.  On error both lengths are zeroed, except QZ_BUF_ERROR which
 * reports the whole chunks / members that did fit (callers loop on it). */
int qzCompress(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
               unsigned int *dest_len, unsigned int last);
int qzCompressExt(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                  unsigned int *dest_len, unsigned int last, uint64_t *ext_rc);
int qzCompressCrc(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                  unsigned int *dest_len, unsigned int last, unsigned long *crc);
int qzCompressCrcExt(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                     unsigned int *dest_len, unsigned int last, unsigned long *crc, uint64_t *ext_rc);
int qzDecompress(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                 unsigned int *dest_len);
int qzDecompressExt(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                    unsigned int *dest_len, uint64_t *ext_rc);
int qzDecompressCrc(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                    unsigned int *dest_len, unsigned long *crc);
int qzDecompressCrcExt(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                       unsigned int *dest_len, unsigned long *crc, uint64_t *ext_rc);

#define QZ_SKID_PAD_SZ 48
#define QZ_COMPRESSED_SZ_OF_EMPTY_FILE 34
unsigned int qzMaxCompressedLength(unsigned int src_sz, QzSession_T *sess);   /* src/qatzip.c:3022-3068 */

/* ---- process-wide defaults: src/qatzip.c:2780-2928 ---- */
int qzSetDefaults(QzSessionParams_T *defaults);
int qzSetDefaultsDeflate(QzSessionParamsDeflate_T *defaults);
int qzSetDefaultsLZ4(QzSessionParamsLZ4_T *defaults);
int qzSetDefaultsLZ4S(QzSessionParamsLZ4S_T *defaults);
int qzSetDefaultsDeflateExt(QzSessionParamsDeflateExt_T *defaults);
int qzGetDefaults(QzSessionParams_T *defaults);
int qzGetDefaultsDeflate(QzSessionParamsDeflate_T *defaults);
int qzGetDefaultsLZ4(QzSessionParamsLZ4_T *defaults);
int qzGetDefaultsLZ4S(QzSessionParamsLZ4S_T *defaults);
int qzGetDefaultsDeflateExt(QzSessionParamsDeflateExt_T *defaults);

/* ---- pinned memory: src/qatzip_mem.c:102-241 (here: hipHostMalloc) ---- */
void *qzMalloc(size_t sz, int numa, int force_pinned);
void qzFree(void *m);
int qzMemFindAddr(unsigned char *a);

/* ---- streaming: src/qatzip_stream.c:403-781 ---- */
typedef struct QzStream_S {
    unsigned int in_sz;
    unsigned int out_sz;
    unsigned char *in;
    unsigned char *out;
    unsigned int pending_in;
    unsigned int pending_out;
    QzCrcType_T crc_type;
    unsigned int crc_32;
    unsigned long long reserved;
    void *opaque;
} QzStream_T;
int qzCompressStream(QzSession_T *sess, QzStream_T *strm, unsigned int last);
int qzDecompressStream(QzSession_T *sess, QzStream_T *strm, unsigned int last);
int qzEndStream(QzSession_T *sess, QzStream_T *strm);

/* ---- declared by the reference, QAT-silicon specific or without an implementation in the reference
 *      snapshot; exported for link compatibility, return QZ_NOT_SUPPORTED ---- */
int qzCompress2(QzSession_T *sess, const unsigned char *src, unsigned char *dest, qzAsyncCallbackFn callback,
                QzResult_T *qzResults);
int qzDecompress2(QzSession_T *sess, const unsigned char *src, unsigned char *dest, qzAsyncCallbackFn callback,
                  QzResult_T *qzResults);
int qzCompressCrc64(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                    unsigned int *dest_len, unsigned int last, uint64_t *crc);
int qzCompressCrc64Ext(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                       unsigned int *dest_len, unsigned int last, uint64_t *crc, uint64_t *ext_rc);
int qzDecompressCrc64(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                      unsigned int *dest_len, uint64_t *crc);
int qzDecompressCrc64Ext(QzSession_T *sess, const unsigned char *src, unsigned int *src_len, unsigned char *dest,
                         unsigned int *dest_len, uint64_t *crc, uint64_t *ext_rc);
int qzCompressWithMetadataExt(QzSession_T *sess, const unsigned char *src, unsigned int *src_len,
                              unsigned char *dest, unsigned int *dest_len, unsigned int last, uint64_t *ext_rc,
                              QzMetadataBlob_T metadata, uint32_t hw_buff_sz_override, uint32_t comp_thrshold);
int qzDecompressWithMetadataExt(QzSession_T *sess, const unsigned char *src, unsigned int *src_len,
                                unsigned char *dest, unsigned int *dest_len, uint64_t *ext_rc,
                                QzMetadataBlob_T metadata, uint32_t hw_buff_sz_override);
int qzAllocateMetadata(QzMetadataBlob_T *metadata, size_t data_size, uint32_t hw_buff_sz);
int qzFreeMetadata(QzMetadataBlob_T metadata);
int qzMetadataBlockRead(uint32_t block_num, QzMetadataBlob_T metadata, uint32_t *block_offset,
                        uint32_t *block_size, uint32_t *block_flags, uint32_t *block_hash);
int qzMetadataBlockWrite(uint32_t block_num, QzMetadataBlob_T metadata, uint32_t *block_offset,
                         uint32_t *block_size, uint32_t *block_flags, uint32_t *block_hash);
int qzMetadataBlockGetCrc64(uint32_t block_num, QzMetadataBlob_T metadata, uint64_t *input_crc, uint64_t *output_crc);
int qzMetadataBlockGetCrc32(uint32_t block_num, QzMetadataBlob_T metadata, uint32_t *input_crc, uint32_t *output_crc);
int qzGetSessionCrc64Config(QzSession_T *sess, QzCrc64Config_T *crc64_config);
int qzGetSessionCrc32Config(QzSession_T *sess, QzCrc32Config_T *crc32_config);
int qzSetSessionCrc64Config(QzSession_T *sess, QzCrc64Config_T *crc64_config);
int qzSetSessionCrc32Config(QzSession_T *sess, QzCrc32Config_T *crc32_config);
int qzGetSoftwareComponentVersionList(QzSoftwareVersionInfo_T *api_info, unsigned int *num_elem);
int qzGetSoftwareComponentCount(unsigned int *num_elem);

#ifdef __cplusplus
}
#endif
#endif
