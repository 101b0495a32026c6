/*
 * sim_driver.cpp — runs the repo's HIP kernel bodies on the CPU SIMT emulator
 * (hipsim.h) and exposes them to the Python tests through a C ABI.
 * TEST INFRASTRUCTURE: used by `-m "not gpu"` tests to fuzz kernel logic against
 * the oracle in a container without a GPU.  Never part of the product path.
 */
#define QZ_SIM 1
#include "hipsim.h"
#include "../../qatzip_amd/csrc/qzk_deflate_lz77.h"
#include "../../qatzip_amd/csrc/qzk_deflate_huff.h"
#include "../../qatzip_amd/csrc/qzk_deflate_wide.h"
#include "../../qatzip_amd/csrc/qzk_deflate_lz77_lane.h"
#include "../../qatzip_amd/csrc/qzk_inflate.h"
#include "../../qatzip_amd/csrc/qzk_inflate_lane.h"
#include "../../qatzip_amd/csrc/qzk_inflate_spec.h"
#include "../../qatzip_amd/csrc/qzk_checksum.h"
#include "../../qatzip_amd/csrc/qzk_deflate_lazy.h"
#include "../../qatzip_amd/csrc/qzk_lz4.h"
#include <vector>

extern "C" {

/* K1 through its persistent pull kernel: `wgs` workgroups of QZK_K1_WAVES waves share the chunk counter.  The emulator
 * runs the workgroups one after the other and the waves of a workgroup interleaved, so the waves of the first workgroup
 * take every chunk between them, each reusing its column of the (dirty, epoch-tagged) table chunk after chunk - the
 * interesting case.  The table starts out as garbage with a foreign epoch, and the epochs continue from run to run. */
static uint32_t g_k1_epoch = 1;
static void run_k1(uint32_t wgs, const uint8_t *src, uint64_t n, uint32_t chunk_sz, uint32_t nchunks,
                   uint8_t *lc, uint16_t *dist, qzk_lzmeta *meta, const uint32_t *cdesc = nullptr, uint32_t waves = 3,
                   uint8_t *slots = nullptr, uint32_t stride = 0, uint32_t final_chunk = ~0u, uint32_t *olen = nullptr,
                   uint32_t *ocrc = nullptr)
{
    static std::vector<qzk_bkt> tables;
    const size_t need = (size_t)QZK_K1_TABROWS(wgs) * QZK_HSIZE * QZK_K1_TABW;
    if (tables.size() < need) {
        qzk_bkt junk; junk.w0 = 0xabcdabcdu; junk.w1 = 0x12345678u; junk.w2 = 0x9abcdef0u; junk.ep = 0;
        tables.assign(need, junk);
    }
    uint32_t counter = 0;
    sim::launch(wgs, 64 * waves, 0, [&] { qzk_lz77_pull_kernel(src, n, chunk_sz, nchunks, lc, dist, meta, tables.data(), &counter, cdesc, g_k1_epoch,
                                                                slots, stride, final_chunk, olen, ocrc, (const uint32_t *)nullptr, qzk_outp{}); });
    g_k1_epoch += nchunks;
}

/* K1: [0] windows that asked the table themselves, [1] all windows - since the library was loaded */
void sim_k1_counts(unsigned long *out) { out[0] = qzk_sim_count[0]; out[1] = qzk_sim_count[1]; }

/* K1 only: symbols + meta of every chunk */
int sim_lz77(const uint8_t *src, uint64_t n, uint32_t chunk_sz, uint8_t *lc, uint16_t *dist, qzk_lzmeta *meta)
{
    uint32_t nchunks = n ? (uint32_t)((n + chunk_sz - 1) / chunk_sz) : 1;
    run_k1(2, src, n, chunk_sz, nchunks, lc, dist, meta);
    return (int)nchunks;
}

/* K1 + K2: raw deflate stream of all chunks (last: final chunk carries BFINAL) */
static int deflate_variant(int variant, const uint8_t *src, uint64_t n, uint32_t chunk_sz, int last, uint8_t *out,
                           uint64_t *out_len, uint32_t *crcs)
{
    uint32_t nchunks = n ? (uint32_t)((n + chunk_sz - 1) / chunk_sz) : 1;
    /* fused: one chunk's worth of symbols per wave of the two workgroups, whatever the number of chunks */
    const size_t symn = variant == 1 ? (size_t)2 * QZK_K1_WAVES * chunk_sz + 64 : n + 64;
    std::vector<uint8_t> lc(symn);
    std::vector<uint16_t> dist(symn);
    std::vector<qzk_lzmeta> meta(nchunks);
    uint32_t stride = (chunk_sz * 9u / 8u + 1024u + 3u) & ~3u;
    std::vector<uint8_t> slots((size_t)nchunks * stride);
    std::vector<uint32_t> olen(nchunks), ocrc(nchunks);
    if (variant == 1)           /* the product's shape: the wave that parsed a chunk codes it too, in the same LDS */
        run_k1(2, src, n, chunk_sz, nchunks, lc.data(), dist.data(), meta.data(), nullptr, 3, slots.data(), stride,
               last ? nchunks - 1 : ~0u, olen.data(), ocrc.data());
    else {
        run_k1(2, src, n, chunk_sz, nchunks, lc.data(), dist.data(), meta.data());
        sim::launch(nchunks, QZK_HW, 0, [&] {
            qzk_huff_kernel(src, n, chunk_sz, nchunks, lc.data(), dist.data(), meta.data(), slots.data(), stride,
                            last ? nchunks - 1 : ~0u, olen.data(), nullptr);
        });
    }
    if (variant != 1) sim::launch(nchunks, QZK_HT, 0, [&] { qzk_crc_chunks_kernel(src, n, chunk_sz, nchunks, ocrc.data(), nullptr); });
    uint64_t pos = 0;
    for (uint32_t c = 0; c < nchunks; c++) {
        memcpy(out + pos, slots.data() + (size_t)c * stride, olen[c]);
        pos += olen[c];
        if (crcs) crcs[c] = ocrc[c];
    }
    *out_len = pos;
    return (int)nchunks;
}

/* K1w (experimental wide-window parse, one chunk of at most 64 KB per 1024-thread workgroup) + K2: the same stream */
int sim_deflate_wide(const uint8_t *src, uint64_t n, uint32_t chunk_sz, int last, uint8_t *out, uint64_t *out_len, uint32_t *crcs)
{
    uint32_t nchunks = n ? (uint32_t)((n + chunk_sz - 1) / chunk_sz) : 1;
    std::vector<uint8_t> lc(n + 64);
    std::vector<uint16_t> dist(n + 64);
    std::vector<qzk_lzmeta> meta(nchunks);
    uint32_t stride = (chunk_sz * 9u / 8u + 1024u + 3u) & ~3u;
    std::vector<uint8_t> slots((size_t)nchunks * stride);
    std::vector<uint32_t> olen(nchunks), ocrc(nchunks);
    std::vector<uint16_t> prevtab((size_t)2 * 65536, 0x5a5a);
    uint32_t counter = 0;
    sim::launch(2, QZX_W, 0, [&] { qzk_lz77_wide_kernel(src, n, chunk_sz, nchunks, lc.data(), dist.data(), meta.data(), prevtab.data(), &counter, nullptr, nullptr); });
    sim::launch(nchunks, QZK_HW, 0, [&] {
        qzk_huff_kernel(src, n, chunk_sz, nchunks, lc.data(), dist.data(), meta.data(), slots.data(), stride,
                        last ? nchunks - 1 : ~0u, olen.data(), nullptr);
    });
    sim::launch(nchunks, QZK_HT, 0, [&] { qzk_crc_chunks_kernel(src, n, chunk_sz, nchunks, ocrc.data(), nullptr); });
    uint64_t pos = 0;
    for (uint32_t c = 0; c < nchunks; c++) {
        memcpy(out + pos, slots.data() + (size_t)c * stride, olen[c]);
        pos += olen[c];
        if (crcs) crcs[c] = ocrc[c];
    }
    *out_len = pos;
    return (int)nchunks;
}

int sim_deflate(const uint8_t *src, uint64_t n, uint32_t chunk_sz, int last, uint8_t *out, uint64_t *out_len, uint32_t *crcs)
{
    return deflate_variant(0, src, n, chunk_sz, last, out, out_len, crcs);
}

int sim_deflate_fused(const uint8_t *src, uint64_t n, uint32_t chunk_sz, int last, uint8_t *out, uint64_t *out_len, uint32_t *crcs)
{
    return deflate_variant(1, src, n, chunk_sz, last, out, out_len, crcs);
}

/* K1b (one chunk per lane) + K2 */
static int deflate_lane_level(const uint8_t *src, uint64_t n, uint32_t chunk_sz, int last, int level, uint8_t *out,
                              uint64_t *out_len, uint32_t *crcs)
{
    uint32_t nchunks = n ? (uint32_t)((n + chunk_sz - 1) / chunk_sz) : 1;
    std::vector<uint8_t> lc(n + 64);
    std::vector<uint16_t> dist(n + 64);
    std::vector<qzk_lzmeta> meta(nchunks);
    uint32_t stride = (chunk_sz * 9u / 8u + 1024u + 3u) & ~3u;
    std::vector<uint8_t> slots((size_t)nchunks * stride);
    std::vector<uint32_t> olen(nchunks), ocrc(nchunks);
    std::vector<uint16_t> head((size_t)nchunks * QZK_HSIZE, 0), prev((size_t)nchunks * QZK_WSIZE, 0x5a5a);
    sim::launch((nchunks + 63) / 64, 64, 0, [&] {
        qzk_lz77_lane_kernel(src, n, chunk_sz, nchunks, lc.data(), dist.data(), meta.data(), head.data(), prev.data(),
                             qzk_level_cfg(level), nullptr);
    });
    sim::launch(nchunks, QZK_HW, 0, [&] {
        qzk_huff_kernel(src, n, chunk_sz, nchunks, lc.data(), dist.data(), meta.data(), slots.data(), stride,
                        last ? nchunks - 1 : ~0u, olen.data(), nullptr);
    });
    sim::launch(nchunks, QZK_HT, 0, [&] { qzk_crc_chunks_kernel(src, n, chunk_sz, nchunks, ocrc.data(), nullptr); });
    uint64_t pos = 0;
    for (uint32_t c = 0; c < nchunks; c++) {
        memcpy(out + pos, slots.data() + (size_t)c * stride, olen[c]);
        pos += olen[c];
        if (crcs) crcs[c] = ocrc[c];
    }
    *out_len = pos;
    return (int)nchunks;
}

int sim_deflate_lane(const uint8_t *src, uint64_t n, uint32_t chunk_sz, int last, uint8_t *out, uint64_t *out_len, uint32_t *crcs)
{
    return deflate_lane_level(src, n, chunk_sz, last, 1, out, out_len, crcs);
}

/* K1b at any zlib level (2-3 greedy, 4-9 lazy) + K2 */
int sim_deflate_level(const uint8_t *src, uint64_t n, uint32_t chunk_sz, int last, int level, uint8_t *out, uint64_t *out_len,
                      uint32_t *crcs)
{
    return deflate_lane_level(src, n, chunk_sz, last, level, out, out_len, crcs);
}

/* a coalesced launch: nchunks slots of chunk_sz bytes, slot k holds cdesc[k] & 0x7fffffff bytes and closes its request's
 * stream when bit 31 is set.  Output: every slot's deflate bytes back to back, per-slot lengths and CRCs. */
int sim_deflate_ragged(const uint8_t *src, uint32_t nchunks, uint32_t chunk_sz, const uint32_t *cdesc, int level, uint8_t *out,
                       uint32_t *lens, uint32_t *crcs)
{
    const uint64_t n = (uint64_t)nchunks * chunk_sz;
    std::vector<uint8_t> lc(n + 64);
    std::vector<uint16_t> dist(n + 64);
    std::vector<qzk_lzmeta> meta(nchunks);
    uint32_t stride = (chunk_sz * 9u / 8u + 1024u + 3u) & ~3u;
    std::vector<uint8_t> slots((size_t)nchunks * stride);
    std::vector<uint32_t> ocrc(nchunks);
    if (level == 1) run_k1(2, src, n, chunk_sz, nchunks, lc.data(), dist.data(), meta.data(), cdesc);
    else if (level >= 4) {                                  /* the lazy kernels, as the product path routes these levels */
        std::vector<uint32_t> head((size_t)nchunks * QZK_HSIZE, 0);
        std::vector<qzk_lazyrec> rec(n + 64);
        std::vector<qzk_lazyres> res(n + 64);
        const qzk_lvlcfg cfg = qzk_level_cfg(level);
        sim::launch(nchunks, 64, 0, [&] { qzk_lazy_chain_kernel(src, n, chunk_sz, nchunks, cdesc, head.data(), rec.data()); });
        sim::launch(nchunks, 64, 0, [&] { qzk_lazy_search_kernel(src, n, chunk_sz, nchunks, cdesc, rec.data(), res.data(), cfg); });
        sim::launch(nchunks, 64, 0, [&] {
            qzk_lazy_parse_kernel(src, n, chunk_sz, nchunks, cdesc, res.data(), lc.data(), dist.data(), meta.data(), cfg);
        });
    } else {
        std::vector<uint16_t> head((size_t)nchunks * QZK_HSIZE, 0), prev((size_t)nchunks * QZK_WSIZE, 0x5a5a);
        sim::launch((nchunks + 63) / 64, 64, 0, [&] {
            qzk_lz77_lane_kernel(src, n, chunk_sz, nchunks, lc.data(), dist.data(), meta.data(), head.data(), prev.data(),
                                 qzk_level_cfg(level), cdesc);
        });
    }
    sim::launch(nchunks, QZK_HW, 0, [&] {
        qzk_huff_kernel(src, n, chunk_sz, nchunks, lc.data(), dist.data(), meta.data(), slots.data(), stride, ~0u, lens, cdesc);
    });
    sim::launch(nchunks, QZK_HT, 0, [&] { qzk_crc_chunks_kernel(src, n, chunk_sz, nchunks, ocrc.data(), cdesc); });
    uint64_t pos = 0;
    for (uint32_t c = 0; c < nchunks; c++) {
        memcpy(out + pos, slots.data() + (size_t)c * stride, lens[c]);
        pos += lens[c];
        crcs[c] = ocrc[c];
    }
    return (int)pos;
}

/* comp_lvl 4-9 through the three lazy kernels (chains, parallel searches, serial parse) + K2 */
int sim_deflate_lazy(const uint8_t *src, uint64_t n, uint32_t chunk_sz, int last, int level, uint8_t *out, uint64_t *out_len,
                     uint32_t *crcs)
{
    uint32_t nchunks = n ? (uint32_t)((n + chunk_sz - 1) / chunk_sz) : 1;
    const size_t span = (size_t)nchunks * chunk_sz;
    std::vector<uint8_t> lc(span + 64);
    std::vector<uint16_t> dist(span + 64);
    std::vector<qzk_lazyrec> pd(span + 64);
    std::vector<qzk_lazyres> res(span + 64);
    std::vector<qzk_lzmeta> meta(nchunks);
    uint32_t stride = (chunk_sz * 9u / 8u + 1024u + 3u) & ~3u;
    std::vector<uint8_t> slots((size_t)nchunks * stride);
    std::vector<uint32_t> olen(nchunks), ocrc(nchunks);
    std::vector<uint32_t> head((size_t)nchunks * QZK_HSIZE, 0);
    const qzk_lvlcfg cfg = qzk_level_cfg(level);
    sim::launch(nchunks, 64, 0, [&] { qzk_lazy_chain_kernel(src, n, chunk_sz, nchunks, nullptr, head.data(), pd.data()); });
    sim::launch(nchunks, 64, 0, [&] { qzk_lazy_search_kernel(src, n, chunk_sz, nchunks, nullptr, pd.data(), res.data(), cfg); });
    sim::launch(nchunks, 64, 0, [&] {
        qzk_lazy_parse_kernel(src, n, chunk_sz, nchunks, nullptr, res.data(), lc.data(), dist.data(), meta.data(), cfg);
    });
    sim::launch(nchunks, QZK_HW, 0, [&] {
        qzk_huff_kernel(src, n, chunk_sz, nchunks, lc.data(), dist.data(), meta.data(), slots.data(), stride,
                        last ? nchunks - 1 : ~0u, olen.data(), nullptr);
    });
    sim::launch(nchunks, QZK_HT, 0, [&] { qzk_crc_chunks_kernel(src, n, chunk_sz, nchunks, ocrc.data(), nullptr); });
    uint64_t pos = 0;
    for (uint32_t c = 0; c < nchunks; c++) {
        memcpy(out + pos, slots.data() + (size_t)c * stride, olen[c]);
        pos += olen[c];
        if (crcs) crcs[c] = ocrc[c];
    }
    *out_len = pos;
    return (int)nchunks;
}

unsigned sim_meta_size(void) { return (unsigned)sizeof(qzk_lzmeta); }

/* K3: inflate nsegs segments described by (in_off, out_off, in_len, out_cap, flags, pad) records */
int sim_inflate(const uint8_t *comp, uint8_t *out, const qzk_infseg *segs, qzk_infres *res, uint32_t nsegs)
{
    uint32_t grid = (nsegs + QZK_INF_WAVES - 1) / QZK_INF_WAVES;
    sim::launch(grid, 64 * QZK_INF_WAVES, 0, [&] { qzk_inflate_kernel(comp, out, segs, res, nsegs); });
    return 0;
}

/* K4: nframes LZ4 frames of frame_sz content bytes each -> slots (stride bytes apart) + lengths */
int sim_lz4c(const uint8_t *src, uint64_t n, uint32_t frame_sz, uint8_t *slots, uint32_t stride, uint32_t *out_len)
{
    uint32_t nframes = n ? (uint32_t)((n + frame_sz - 1) / frame_sz) : 1;
    sim::launch(nframes, 64, 0, [&] { qzk_lz4c_kernel(src, n, frame_sz, nframes, slots, stride, out_len, 0); });
    return (int)nframes;
}
/* the same frames by persistent waves that pull frame numbers, hash tables outside LDS (qzk_lz4c_pull_kernel): fewer waves than frames */
int sim_lz4c_pull(const uint8_t *src, uint64_t n, uint32_t frame_sz, uint8_t *slots, uint32_t stride, uint32_t *out_len, uint32_t waves)
{
    uint32_t nframes = n ? (uint32_t)((n + frame_sz - 1) / frame_sz) : 1;
    /* the waves' tables and epochs live on from launch to launch, as on the device (cleared once); QZSIM_LZ4_EPOCH0 starts the
     * epochs close to the 16-bit wrap so that the re-clearing is exercised */
    static std::vector<uint64_t> tables;
    static std::vector<uint32_t> epochs;
    if (tables.size() < (size_t)waves * QZK_L4C_TABW) {
        tables.assign((size_t)waves * QZK_L4C_TABW, 0ull);
        epochs.assign(waves, getenv("QZSIM_LZ4_EPOCH0") ? (uint32_t)atoi(getenv("QZSIM_LZ4_EPOCH0")) : 0u);
    }
    uint32_t counter = 0;
    sim::launch(waves, 64, 0, [&] { qzk_lz4c_pull_kernel(src, n, frame_sz, nframes, slots, stride, out_len, 0, tables.data(), epochs.data(), &counter); });
    return (int)nframes;
}
/* the same frames behind the hardware path's header (FLG 0x4C, content size always there) */
int sim_lz4c_hw(const uint8_t *src, uint64_t n, uint32_t frame_sz, uint8_t *slots, uint32_t stride, uint32_t *out_len)
{
    uint32_t nframes = n ? (uint32_t)((n + frame_sz - 1) / frame_sz) : 1;
    sim::launch(nframes, 64, 0, [&] { qzk_lz4c_kernel(src, n, frame_sz, nframes, slots, stride, out_len, 1); });
    return (int)nframes;
}

/* K4, the hardware framing's chunks above 64 KB: a linked frame per chunk, one launch */
int sim_lz4c_linked_many(const uint8_t *src, uint64_t total, uint32_t chunk, uint32_t nfr, uint8_t *slots, uint32_t stride, uint32_t *lens)
{
    sim::launch(nfr, 64, 0, [&] { qzk_lz4c_linked_many_kernel(src, total, chunk, nfr, slots, stride, lens); });
    return 0;
}

/* K4, linked mode: ONE frame for a call above 64 KB */
int sim_lz4c_linked(const uint8_t *src, uint32_t n, uint8_t *out, uint32_t *out_len)
{
    sim::launch(1, 64, 0, [&] { qzk_lz4c_linked_kernel(src, n, out, out_len); });
    return 0;
}

/* K5: decode nsegs frames */
int sim_lz4d(const uint8_t *comp, uint8_t *out, const qzk_lz4seg *segs, qzk_lz4res *res, uint32_t nsegs)
{
    sim::launch(nsegs, 64, 0, [&] { qzk_lz4d_kernel(comp, out, segs, res, nsegs); });
    return 0;
}

/* K3b, serial phase A (one segment per lane -> literal streams + sequence records) + phase B (one wave per segment) */
static void two_phase_serial(const uint8_t *comp, uint8_t *out, const qzk_infseg *segs, qzk_infres *res, uint32_t nsegs)
{
    std::vector<qzk_inf_tab> tabs(nsegs);
    std::vector<qzk_tokseg> ts(nsegs);
    std::vector<qzk_chain> chains(nsegs);
    /* one arena, a region per segment: literals up from its first byte, sequences down from its end (qzk_inflate_lane.h) */
    uint64_t lt = 0;
    for (uint32_t i = 0; i < nsegs; i++) {
        const uint64_t rg = segs[i].flags & QZK_INF_COUNT_ONLY ? 0 : QZK_TOK_REGION(segs[i].out_cap);
        ts[i].lit_off = lt; ts[i].seq_off = (lt + rg) / 8;
        lt += rg;
    }
    std::vector<uint64_t> arena((lt + 1088) / 8 + 1, 0xeeeeeeeeeeeeeeeeull);          /* phase B reads whole 16-byte rows (qzk_lz_batch.h) */
    uint8_t *const lits_p = (uint8_t *)arena.data(); qzk_seq *const seqs_p = (qzk_seq *)arena.data();
    /* 16 segments per workgroup; the emulator wants whole waves, the kernel bounds-checks */
    sim::launch((nsegs + 15) / 16, 64, 0, [&] {
        if (threadIdx.x < 16) qzk_inflate_tok_kernel<16>(comp, segs, res, nsegs, tabs.data(), ts.data(), lits_p, seqs_p, chains.data(), nullptr, 0u, 1u);
    });
    /* phase B the way the host streams output: launches over index lists (here: the segments in reverse, two parts) */
    std::vector<uint32_t> ord(nsegs);
    for (uint32_t i = 0; i < nsegs; i++) ord[i] = nsegs - 1 - i;
    const uint32_t half = nsegs / 2;
    for (int part = 0; part < 2; part++) {
        const uint32_t first = part ? half : 0, cnt = part ? nsegs - half : half;
        if (!cnt) continue;
        sim::launch((cnt + QZK_RES_WAVES - 1) / QZK_RES_WAVES, 64 * QZK_RES_WAVES, 0, [&] {
            qzk_lz_resolve_kernel(comp, out, segs, res, nsegs, ts.data(), 1u, lits_p, seqs_p, chains.data(),
                                  ord.data() + first, cnt);
        });
    }
}

int sim_inflate_lane(const uint8_t *comp, uint8_t *out, const qzk_infseg *segs, qzk_infres *res, uint32_t nsegs)
{
    two_phase_serial(comp, out, segs, res, nsegs);
    return 0;
}

} /* extern "C" */

/* K3b with the speculative phase A (K lanes per segment); segments it hands back are redone serially, like the host
 * does.  Returns how many segments were handed back. */
template <int K>
static int two_phase_spec(const uint8_t *comp, uint8_t *out, const qzk_infseg *segs, qzk_infres *res, uint32_t nsegs)
{
    std::vector<qzk_inf_tab> tabs(nsegs);
    std::vector<qzk_tokseg> ts((size_t)nsegs * K);
    std::vector<qzk_chain> chains(nsegs);
    std::vector<qzk_rec> recs((size_t)nsegs * K * QZK_SPEC_NREC);
    uint64_t lt = 0;
    for (uint32_t i = 0; i < nsegs; i++)
        for (int j = 0; j < K; j++) {
            const uint64_t rg = QZK_SPEC_REGION(segs[i].out_cap, K, j);
            ts[(size_t)i * K + j].lit_off = lt; ts[(size_t)i * K + j].seq_off = (lt + rg) / 8;
            lt += rg;
        }
    std::vector<uint64_t> arena((lt + 1088) / 8 + 1, 0xeeeeeeeeeeeeeeeeull);          /* phase B reads whole 16-byte rows (qzk_lz_batch.h) */
    uint8_t *const lits_p = (uint8_t *)arena.data(); qzk_seq *const seqs_p = (qzk_seq *)arena.data();
    const uint32_t spw = 64 / K;
    sim::launch((nsegs + spw - 1) / spw, 64, 0, [&] {
        static uint64_t epoch = (5ull << 22) - 40;               /* launch numbers that cross the tag's own 22 bits: the high part lives in the record's second word */
        if (threadIdx.x == 0 && blockIdx.x == 0) epoch++;
        qzk_inflate_spec_kernel<K, 2>(comp, segs, res, nsegs, tabs.data(), ts.data(), lits_p, seqs_p, chains.data(), recs.data(), epoch + 1, (uint32_t)(getenv("QZSIM_OVER") ? atoi(getenv("QZSIM_OVER")) : 1));
    });
    sim::launch((nsegs + QZK_RES_WAVES - 1) / QZK_RES_WAVES, 64 * QZK_RES_WAVES, 0, [&] {
        qzk_lz_resolve_kernel(comp, out, segs, res, nsegs, ts.data(), (uint32_t)K, lits_p, seqs_p, chains.data(), nullptr, 0);
    });
    if (getenv("QZSIM_TRACE"))
        for (uint32_t i = 0; i < nsegs; i++) {
            fprintf(stderr, "seg %u status %d pieces %u:", i, res[i].status, chains[i].nel);
            for (uint32_t e = 0; e < chains[i].nel; e++) fprintf(stderr, " [sub %u seq %u+%u]", chains[i].el[e].sub, chains[i].el[e].seq_first, chains[i].el[e].seq_count);
            fprintf(stderr, "\n");
        }
    std::vector<uint32_t> redo;
    for (uint32_t i = 0; i < nsegs; i++) if (res[i].status == QZK_INF_ESPEC) redo.push_back(i);
    if (!redo.empty()) {
        std::vector<qzk_infseg> rs(redo.size());
        std::vector<qzk_infres> rr(redo.size());
        for (size_t i = 0; i < redo.size(); i++) rs[i] = segs[redo[i]];
        two_phase_serial(comp, out, rs.data(), rr.data(), (uint32_t)redo.size());
        for (size_t i = 0; i < redo.size(); i++) res[redo[i]] = rr[i];
    }
    return (int)redo.size();
}

extern "C" {

#ifdef QZK_SPEC_STATS
uint32_t *sim_spec_stats() { return qzk_spec_stats; }
#endif
int sim_inflate_spec(const uint8_t *comp, uint8_t *out, const qzk_infseg *segs, qzk_infres *res, uint32_t nsegs, int K)
{
    if (K == 2) return two_phase_spec<2>(comp, out, segs, res, nsegs);
    if (K == 4) return two_phase_spec<4>(comp, out, segs, res, nsegs);
    if (K == 16) return two_phase_spec<16>(comp, out, segs, res, nsegs);
    if (K == 32) return two_phase_spec<32>(comp, out, segs, res, nsegs);
    return two_phase_spec<8>(comp, out, segs, res, nsegs);
}

int sim_adler(const uint8_t *data, uint64_t n, uint32_t chunk_sz, uint32_t *out)
{
    uint32_t nchunks = n ? (uint32_t)((n + chunk_sz - 1) / chunk_sz) : 1;
    sim::launch(nchunks, QZK_HT, 0, [&] { qzk_adler_chunks_kernel(data, n, chunk_sz, nchunks, out); });
    return (int)nchunks;
}

int sim_crc(const uint8_t *data, const qzk_range *ranges, uint32_t nranges, uint32_t *crc_out)
{
    sim::launch(nranges, QZK_HT, 0, [&] { qzk_crc_kernel(data, ranges, nranges, crc_out); });
    return 0;
}
}
