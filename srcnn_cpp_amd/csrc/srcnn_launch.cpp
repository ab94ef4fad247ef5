// srcnn_launch.cpp -- run_strip(): the ONE launch path of the MFMA strip kernels (fused, layers 1-2, layer 3; float32 and
// split-f16), their seam launches and the SRCNN_MODE_REFBYTES fix-up behind them; and the device-pointer entry points of
// include/srcnn_amd.h that are thin argument checks over it.
#include "srcnn_ctx.h"

using namespace srcnn;
using namespace srcnn::host;

namespace srcnn {
namespace host {

// Do two element ranges of the same device address space overlap?  (first byte, one past the last byte)
bool ranges_overlap(const void *a, size_t a_bytes, const void *b, size_t b_bytes)
{
    const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
    return a0 < b0 + b_bytes && b0 < a0 + a_bytes;
}
size_t span_elems(size_t stride, size_t frame_pitch, int width, int height, int n_frames)
{
    return (size_t)(n_frames - 1) * frame_pitch + (size_t)(height - 1) * stride + (size_t)width;
}

// (row strides stay below 2^30 elements: the kernels add a lane's column to one row stride in 32 bits)
bool bad_plane(const void *p, size_t stride, int w, int h)
{
    return !p || w <= 0 || h <= 0 || stride < (size_t)w || stride >= ((size_t)1 << 30);
}
// the kernels address a lane's plane element with a 32-bit offset from a per-plane scalar base
bool bad_pitch(size_t plane_pitch) { return plane_pitch >= ((size_t)1 << 29); }

// Seam deferral: queue the seam launch a fused launch left pending (srcnn_ctx::PendingSeams), on the stream it belongs to.
int flush_seams(srcnn_ctx *c)
{
    if (!c || !c->pending.valid) return SRCNN_OK;
    c->pending.valid = false;
    const FoldParams &f = c->pending.f;
    if (f.winmap) HIP_TRY(c, launch_seams_merged(f.prev, f.n_seams, f.seams, f.winmap, 1, c->pending.stream));
    else HIP_TRY(c, launch_seams(f.prev, f.n_seams, f.seams, c->pending.stream));
    return SRCNN_OK;
}

// May the seam blocks of the pending launch `a` run inside launch `b`'s kernel?  They write a's seam pixels while b's work items
// READ b's input and write b's regular pixels.  (1) b must read nothing a's seam blocks write: a chain (b's src, or one of its halo
// buffers, is a's output) would otherwise see a's seam pixels before they are finished.  (2) The two outputs must be disjoint, or b
// the SAME launch again (same output, same geometry, same work-item plan -- a step loop on one output buffer: b's own seam pixels
// are the ones a's blocks write, and b's seam blocks rewrite them behind b).  Anything else that overlaps (another geometry into
// the same buffer) could let a stale seam pixel of a land on a finished pixel of b: queue a's seam launch first.
static bool fold_is_safe(const StripParams &a, const StripParams &b)
{
    auto span = [](const StripParams &q, const uint8_t **lo, size_t *bytes) {
        *lo = q.dst + (long)(q.row_begin - q.dst_row0) * q.dst_stride;
        *bytes = (size_t)(q.row_end - q.row_begin - 1) * (size_t)q.dst_stride + (size_t)q.width;
    };
    const uint8_t *alo, *blo;
    size_t ab, bb;
    span(a, &alo, &ab);
    span(b, &blo, &bb);
    // the input rows b's work items read: [row_begin - 6, row_end + 6) clipped to the plane, wherever they lie (src, or a halo buffer)
    {
        const int r0 = std::max(0, b.row_begin - kHaloRows), r1 = std::min(b.height, b.row_end + kHaloRows);
        const int s0 = b.src_top ? std::max(r0, b.src_row0) : r0, s1 = b.src_bot ? std::min(r1, b.src_row1) : r1;
        if (s1 > s0 && ranges_overlap(alo, ab, b.src + (long)(s0 - b.src_row0) * b.src_stride,
                                      (size_t)(s1 - s0 - 1) * (size_t)b.src_stride + (size_t)b.width))
            return false;
        const size_t halo_bytes = (size_t)(kHaloRows - 1) * (size_t)b.halo_stride + (size_t)b.width;
        if (b.src_top && ranges_overlap(alo, ab, b.src_top, halo_bytes)) return false;
        if (b.src_bot && ranges_overlap(alo, ab, b.src_bot, halo_bytes)) return false;
    }
    if (!ranges_overlap(alo, ab, blo, bb)) return true;
    return a.dst == b.dst && a.dst_stride == b.dst_stride && a.dst_row0 == b.dst_row0 && a.width == b.width && a.height == b.height &&
           a.row_begin == b.row_begin && a.row_end == b.row_end && a.items == b.items && a.strips_total == b.strips_total;
}

// FixParams::kl from StripParams::fix_kl: the split-f16 kernel's factor carries the 2^-10 of its scaled plane 5, the fix-up kernels'
// S1 is in true units
static float r16_kl(const srcnn_ctx *c, float strip_kl) { return c->mode == SRCNN_MODE_REFBYTES16 ? strip_kl * 1024.f : strip_kl; }

// Common launch of the three strip modes on device memory.
// fix_frame / fix_frames (SRCNN_MODE_REFBYTES only): this single-frame launch is frame `fix_frame` of a batch of `fix_frames`
// whose flagged pixels ONE fix-up finishes, queued behind the batch's last launch (fix_frames = 1: the launch's own fix-up).
int run_strip(srcnn_ctx *c, int mode, StripParams p, int n_frames, int fix_frame, int fix_frames, bool may_defer)
{
    const int halo = (mode == MODE_L12) ? 0 : 2;
    // Undocumented experiment knobs (never set in production).  Only bits that leave every output byte as it is are honoured:
    // 2 / 16 = the stamped builds of the split-f16 / float32 production kernel (tools/diag_split16.py, diag_light.py), 8 = no XCD remap,
    // 128 = small batches on the regular grid.  A stray SRCNN_DEBUG_TUNE cannot change a pixel (tests/test_gpu_hardening.py).
    static const char *env_tune = SRCNN_DEBUG_ENV("SRCNN_DEBUG_TUNE");
    static const char *env_pad = SRCNN_DEBUG_ENV("SRCNN_DEBUG_LDS_PAD");
    constexpr int kTuneHarmless = 2 | 8 | 16 | 128;
    p.tune = (env_tune ? std::atoi(env_tune) : 0) & kTuneHarmless;
    const size_t pad = env_pad ? (size_t)std::atol(env_pad) : 0;
    const bool split16 = mode == MODE_FUSED && (c->mode == SRCNN_MODE_SPLIT16 || c->mode == SRCNN_MODE_REFBYTES16);
    const int wgs_per_cu = split16_wgs_per_cu(split16, p.tune);
    Plan pl = make_plan(c, p.width, p.row_end - p.row_begin, n_frames, halo, wgs_per_cu);
    static const char *env_segs = SRCNN_DEBUG_ENV("SRCNN_DEBUG_SEGS");     // experiment knob
    if (env_segs && std::atoi(env_segs) > 0) {
        const int rows = p.row_end - p.row_begin, ns = std::min(rows, std::atoi(env_segs));
        pl.seg_rows = (rows + ns - 1) / ns;
        pl.n_segs = (rows + pl.seg_rows - 1) / pl.seg_rows;
    }
    p.seg_rows = pl.seg_rows;
    p.n_strips = pl.n_strips;
    p.n_segs = pl.n_segs;
    p.items = nullptr;
    p.seam = nullptr;
    p.cseam = nullptr;
    p.strips_total = pl.n_strips;
    int grid_items = 0;
    bool defer = false;                  // seam deferral: leave this launch's seam work pending (see below)
    const srcnn_ctx::ItemTable *table = nullptr;
    // Convolution55 alone (MODE_L3) reads 128 B per pixel and is bound by HBM: strips of exactly FW = 128 columns
    // (column seams instead of 2 halo columns each side), so that the four waves of a workgroup read four whole
    // 128-byte lines per plane and row -- with 124-column strips every strip start falls inside a line and a fifth
    // line is fetched: 1.32 x the algorithmic bytes by FETCH_SIZE (profiles/r02) -- and four workgroups per CU
    // (<= 128 VGPRs, 18 KB of LDS) to keep 64 KB of loads in flight per CU.  SRCNN_DEBUG_L3=0: the round-1 launch.
    static const char *env_l3 = SRCNN_DEBUG_ENV("SRCNN_DEBUG_L3");
    const int ns_l3 = (p.width + FW - 1) / FW;
    const bool l3_aligned = mode == MODE_L3 && !(env_l3 && std::atoi(env_l3) == 0) &&
                            (p.width - (ns_l3 - 1) * FW >= 4 || ns_l3 == 1);
    if (l3_aligned) {
        pl = make_plan(c, p.width, p.row_end - p.row_begin, n_frames, halo, 4, 0);
        p.seg_rows = pl.seg_rows;
        p.n_strips = pl.n_strips;
        p.n_segs = pl.n_segs;
        p.strips_total = pl.n_strips;
        srcnn_ctx::SeamScratch *sc = nullptr;
        int rc;
        if ((rc = seam_scratch_for_stream(c, &sc))) return rc;
        const size_t n = (size_t)n_frames * p.strips_total * (p.row_end - p.row_begin) * CSEAM_FLOATS * sizeof(float);
        if ((rc = reserve(c, sc->cbuf, n))) return rc;
        p.cseam = static_cast<float *>(sc->cbuf.p);
    }
    // One plane that fits the GPU in a single round: size the work items by the speed of the wave
    // slot they will land in and use every slot (build_items).
    const bool fused32 = mode == MODE_FUSED && !split16;
    // A small batch repeats the plane's item plan frame after frame in one launch (no halo rows, and the next frame's
    // blocks fill the CUs the last items of a frame leave idle): 8 x 3840x2160 0.857 against 0.839 on the regular
    // grid; from kItemBatchMax frames on the regular grid's tall segments are as good (64 frames: 0.863 vs 0.865).
    // (srcnn_forward_y_dev hands over at most kItemBatchChunk frames of such a batch per call: frames_per_launch())
    if (mode != MODE_L12 && (n_frames == 1 || (fused32 && n_frames < kItemBatchMax && !(p.tune & 128))) && !l3_aligned) {   // tune 128: regular grid (A/B)
        // float32 fused kernel only: seams instead of halo rows between the items of a strip, and column seams
        // instead of halo columns between strips (srcnn_kernels.h).  SRCNN_DEBUG_SEAMS: 0 = neither, 1 = rows only.
        static const char *env_seams = SRCNN_DEBUG_ENV("SRCNN_DEBUG_SEAMS");
        const int seam_knob = env_seams ? std::atoi(env_seams) : 3;
        const bool want_seams = mode == MODE_FUSED && !split16 && (seam_knob & 1);
        int rc;
        // a plane too small for two items per CU of useful height: one (taller) item per CU still beats the regular grid
        // with its halo rows
        auto items_for = [&](int n_strips_, bool seams_) -> int {
            int rc2 = build_items(c, n_strips_, p.row_begin, p.row_end, p.height, wgs_per_cu, seams_, &table);
            if (!rc2 && table->count == 0 && wgs_per_cu == 2 && seams_)
                rc2 = build_items(c, n_strips_, p.row_begin, p.row_end, p.height, 1, seams_, &table);
            return rc2;
        };
        bool col_seams = false;
        if (want_seams && (seam_knob & 2) && cseam_pays(p.width)) {
            // strips of FW output columns; the last strip must hold the 4 columns its left neighbour's pixels need
            const int ns_cs = (p.width + FW - 1) / FW;
            if ((rc = items_for(ns_cs, true))) return rc;
            if (table->count > 0) {
                p.strips_total = ns_cs;
                col_seams = true;
            } else {
                table = nullptr;
            }
        }
        if (!table && (rc = items_for(pl.n_strips, want_seams))) return rc;
        grid_items = table->count;
        if (grid_items > 0) {
            p.items = static_cast<const int *>(table->dev.p);
            if (table->n_seams > 0 || col_seams) {
                srcnn_ctx::SeamScratch *sc = nullptr;
                for (auto &e : c->seam_scratch)
                    if (e.used && e.stream == c->stream) sc = &e;
                for (auto &e : c->seam_scratch)
                    if (!sc && !e.used) sc = &e;
                if (!sc) {                  // more streams than slots: wait for everything, start over with slot 0
                    HIP_TRY(c, hipDeviceSynchronize());
                    for (auto &e : c->seam_scratch) e.used = false;
                    sc = &c->seam_scratch[0];
                }
                sc->used = true;
                sc->stream = c->stream;
                // Seam deferral: this launch's seam blocks will ride behind the NEXT launch's work items, which writes its own
                // exports meanwhile -- the two scratch sets of the stream are used in turn.
                defer = may_defer && c->defer_seams && c->defer_block == 0 && !c->safe_hazards && fused32 && n_frames == 1 && !p.pre && !(p.tune & 16) &&
                        table->n_seams > 0 && (col_seams ? table->separated : true) &&
                        !(c->mode == SRCNN_MODE_REFBYTES || c->mode == SRCNN_MODE_REFBYTES16);
                if (defer) sc->flip ^= 1;
                DevBuf &rbuf = defer && sc->flip ? sc->buf2 : sc->buf, &cbuf = defer && sc->flip ? sc->cbuf2 : sc->cbuf;
                if (table->n_seams > 0) {
                    if ((rc = reserve(c, rbuf, (size_t)n_frames * table->n_seams * SEAM_FLOATS * NTHREADS * sizeof(float)))) return rc;
                    p.seam = static_cast<float *>(rbuf.p);
                }
                if (col_seams) {
                    const size_t n = (size_t)n_frames * p.strips_total * (p.row_end - p.row_begin) * CSEAM_FLOATS * sizeof(float);
                    if ((rc = reserve(c, cbuf, n))) return rc;
                    p.cseam = static_cast<float *>(cbuf.p);
                }
            }
            p.n_strips = 1;            // grid = n_strips * n_segs * n_frames blocks
            p.n_segs = grid_items;
            p.items_per_frame = grid_items;
            p.seams_per_frame = std::max(1, table->n_seams);
        }
    }
    // Batches (regular grid): column seams only -- the planner already makes the segments tall, and a row seam
    // costs 74 KB of scratch.
    if (fused32 && n_frames > 1 && grid_items == 0) {
        static const char *env_seams = SRCNN_DEBUG_ENV("SRCNN_DEBUG_SEAMS");
        const int ns_cs = (p.width + FW - 1) / FW;
        (void)ns_cs;
        if ((!env_seams || (std::atoi(env_seams) & 2)) && cseam_pays(p.width)) {
            pl = make_plan(c, p.width, p.row_end - p.row_begin, n_frames, halo, wgs_per_cu, 0);
            p.seg_rows = pl.seg_rows;
            p.n_strips = pl.n_strips;
            p.n_segs = pl.n_segs;
            p.strips_total = pl.n_strips;
            srcnn_ctx::SeamScratch *sc = nullptr;
            for (auto &e : c->seam_scratch)
                if (e.used && e.stream == c->stream) sc = &e;
            for (auto &e : c->seam_scratch)
                if (!sc && !e.used) sc = &e;
            if (!sc) {
                HIP_TRY(c, hipDeviceSynchronize());
                for (auto &e : c->seam_scratch) e.used = false;
                sc = &c->seam_scratch[0];
            }
            sc->used = true;
            sc->stream = c->stream;
            const size_t n = (size_t)n_frames * p.strips_total * (p.row_end - p.row_begin) * CSEAM_FLOATS * sizeof(float);
            int rc;
            if ((rc = reserve(c, sc->cbuf, n))) return rc;
            p.cseam = static_cast<float *>(sc->cbuf.p);
        }
    }
    // SRCNN_MODE_REFBYTES: the fused float32 kernel also writes a flag byte per pixel; fix_collect / fix_apply then recompute
    // the flagged pixels in the reference's arithmetic (srcnn_exact.hip).  One frame per strip launch; the fix-up of up to
    // FIX_BATCH_FRAMES consecutive frames of a batch is ONE pair of launches behind the last of them (srcnn_forward_y_dev): its
    // items are drawn from one list, so the draw's tail -- 3.3 rounds of items on a single 3840x2160 plane leave 18 % of the wave
    // slots empty -- is paid once per batch.
    const bool fix = mode == MODE_FUSED && (c->mode == SRCNN_MODE_REFBYTES || c->mode == SRCNN_MODE_REFBYTES16) && !p.pre;
    srcnn_ctx::SeamScratch *fsc = nullptr;
    size_t fix_scat_cap = 0, fix_dense_cap = 0, fix_flag_pitch = 0;
    float fix_delta_used = 0.f;
    if (fix) {
        if (n_frames != 1 || fix_frame < 0 || fix_frame >= fix_frames || fix_frames > FIX_BATCH_FRAMES)
            return fail(c, SRCNN_ERR_STATE, "REFBYTES strip launches hold one frame");
        int rc;
        if ((rc = seam_scratch_for_stream(c, &fsc))) return rc;
        const int rows = p.row_end - p.row_begin;
        fix_scat_cap = fixup_list_entries(p.width, rows, fix_frames, &fix_dense_cap);
        fix_flag_pitch = (size_t)rows * (size_t)p.dst_stride;
        // (sized for the whole batch at its first frame: no buffer moves while earlier frames' flags wait for the fix-up)
        if ((rc = reserve(c, fsc->flag, fix_flag_pitch * (size_t)fix_frames))) return rc;
        if ((rc = reserve(c, fsc->fix_lists, (fix_scat_cap + fix_dense_cap) * sizeof(unsigned)))) return rc;
        if ((rc = reserve(c, fsc->fix_counters, FIX_COUNTERS * sizeof(unsigned)))) return rc;
        if (!c->fix_totals.p) {
            if ((rc = reserve(c, c->fix_totals, FIX_TOTALS * sizeof(unsigned long long)))) return rc;
            HIP_TRY(c, hipMemsetAsync(c->fix_totals.p, 0, FIX_TOTALS * sizeof(unsigned long long), c->stream));
        }
        // flag[o] for the same element offsets o as dst: o >= (row_begin - dst_row0) * dst_stride
        p.flag = static_cast<uint8_t *>(fsc->flag.p) + (size_t)fix_frame * fix_flag_pitch - (long)(p.row_begin - p.dst_row0) * p.dst_stride;
        // (the split-f16 kernel's noise is a little wider than the float32 kernel's -- soak: 4.3e-4 against 3.7e-4 -- and has no
        // CPU model to take statistics from: 8 * E0 instead of 6 * E0, and the same monitor)
        fix_delta_used = c->mode == SRCNN_MODE_REFBYTES16 ? c->fix_delta * (8.f / 6.f) : c->fix_delta;
        p.fix_delta = fix_delta_used;
        p.fix_scale = 253.f / (2.f * fix_delta_used);
        // the per-pixel threshold min(delta, kl * S1 + abs) of both strip kernels (srcnn_kernels.h, l3_row_is_scale());
        // srcnn_set_fixup_local(ctx, 0): the one global threshold of rounds 3-5
        const bool r16 = c->mode == SRCNN_MODE_REFBYTES16;
        const bool local = (r16 ? c->fix_local16 : c->fix_local) > 0.f;
        // (the split-f16 kernel's plane 5 is x 2^10 like its tap partials: the factor carries the 2^-10)
        p.fix_kl = local ? (r16 ? c->fix_local16 * std::ldexp(1.f, -10) : c->fix_local) * c->fix_margin * std::ldexp(1.f, -24) : 0.f;
        p.fix_abs = local ? kFixAbsLocal : fix_delta_used;
        p.fix_counters = static_cast<unsigned *>(fsc->fix_counters.p);
    }
    p.wfrag = static_cast<const float *>(c->wfrag.p);
    p.wfrag16 = static_cast<const uint32_t *>(c->wfrag16.p);
    p.sink = static_cast<float *>(c->sink.p);
    p.b3 = c->b3;
    if (split16) {
        if (!c->split16_ok)
            return fail(c, SRCNN_ERR_STATE, "SRCNN_MODE_SPLIT16: these weights exceed the f16 ranges of the mode "
                                            "(layer maps must stay below 8192 / 16384 for 8-bit input); use SRCNN_MODE_MFMA");
        int rc;
        if ((rc = flush_seams(c))) return rc;
        HIP_TRY(c, launch_split16(p, n_frames, c->stream, pad));
    }
    else if (c->pending.valid && defer && c->pending.stream == c->stream && grid_items > 0 && fold_is_safe(c->pending.f.prev, p)) {
        // the previous launch's seam blocks ride behind this launch's work items (srcnn_strip_fold_kernel)
        c->pending.f.first_block = grid_items;
        c->pending.valid = false;
        HIP_TRY(c, launch_strip_fold(p, c->pending.f, c->stream, pad));
    } else {
        int rc;
        if ((rc = flush_seams(c))) return rc;
        HIP_TRY(c, (c->safe_hazards ? launch_strip_safe : launch_strip)(mode, p, n_frames, c->stream, pad));
    }
    // Row seams and column seams in ONE launch when the plan keeps the seam windows of neighbouring strips apart
    // (plan_items_balanced()): the blocks that finish a row seam then also finish the column-seam pixels of their four
    // rows, the column-seam blocks skip those rows, and neither waits for the other.
    static const char *env_merge = SRCNN_DEBUG_ENV("SRCNN_DEBUG_SEAM_MERGE");      // experiment knob: 0 = two launches
    if (defer && !(env_merge && std::atoi(env_merge) == 0)) {
        FoldParams &f = c->pending.f;
        f.prev = p;
        f.seams = static_cast<const int *>(table->dev_seams.p);
        f.winmap = p.cseam ? static_cast<const unsigned char *>(table->dev_winmap.p) : nullptr;
        f.n_seams = table->n_seams;
        f.cblocks = p.cseam ? (int)(((long)(p.strips_total - 1) * (p.row_end - p.row_begin) + 255) / 256) : 0;
        f.first_block = 0;
        c->pending.stream = c->stream;
        c->pending.valid = true;
    } else if (p.seam && p.cseam && table->separated && !(env_merge && std::atoi(env_merge) == 0)) {
        HIP_TRY(c, launch_seams_merged(p, table->n_seams * n_frames, static_cast<const int *>(table->dev_seams.p),
                                       static_cast<const unsigned char *>(table->dev_winmap.p), n_frames, c->stream));
    } else {
        if (p.seam) HIP_TRY(c, launch_seams(p, table->n_seams * n_frames, static_cast<const int *>(table->dev_seams.p), c->stream));
        if (p.cseam) HIP_TRY(c, launch_cseams(p, n_frames, c->stream));
    }
    if (fix && fix_frame == fix_frames - 1) {
        FixParams f{};
        f.n_frames = fix_frames;                         // frame 0 of the batch lies fix_frame frames before this launch's
        f.src_frame_pitch = p.src_frame_pitch;
        f.dst_frame_pitch = p.dst_frame_pitch;
        f.flag_frame_pitch = (long)fix_flag_pitch;
        f.src = p.src - (long)fix_frame * p.src_frame_pitch;
        f.src_stride = p.src_stride;
        f.src_row0 = p.src_row0;
        f.src_top = p.src_top;
        f.src_bot = p.src_bot;
        f.halo_stride = p.halo_stride;
        f.src_row1 = p.src_row1;
        f.dst = p.dst - (long)fix_frame * p.dst_frame_pitch;
        f.flag = p.flag - (long)fix_frame * (long)fix_flag_pitch;
        f.dst_stride = p.dst_stride;
        f.dst_row0 = p.dst_row0;
        f.width = p.width;
        f.height = p.height;
        f.row_begin = p.row_begin;
        f.row_end = p.row_end;
        f.wraw = static_cast<const float *>(c->wraw.p);
        f.counters = p.fix_counters;
        f.totals = static_cast<unsigned long long *>(c->fix_totals.p);
        f.scat = static_cast<unsigned *>(fsc->fix_lists.p);
        f.dense = f.scat + fix_scat_cap;
        f.delta = fix_delta_used;
        f.code_step = 2.f * fix_delta_used / 253.f;
        f.kl = r16_kl(c, p.fix_kl);          // the fix-up's monitor computes S1 in true units
        f.abs_term = p.fix_abs;
        // The monitor ACTS, on the device (srcnn_set_fixup_strict, on by default): fix_apply_kernel records the largest
        // |v_mfma - v_reference| over the pixels it recomputes -- a random ~0.1-0.3 % sample of the launch -- relative to each
        // pixel's own threshold, and above HALF of it, where the margin the mode rests on is gone for this content / model, fix_rerun_kernel redoes every frame of
        // the fix-up batch in the reference's arithmetic (no threshold involved).  No host read: the stream is never stalled.
        static const char *env_rerun = SRCNN_DEBUG_ENV("SRCNN_DEBUG_FORCE_RERUN");     // test knob: every launch is redone
        f.rerun_above = (env_rerun && std::atoi(env_rerun)) ? -1.f : c->fix_strict ? 0.5f : INFINITY;
        // fix_apply with both weight tables in LDS (3 workgroups per CU) on planes whose items fit ONE round of the draw: an item
        // alone on a compute unit is bound by its scalar weight loads (L2 round trips the scalar cache cannot hold back), which five
        // co-resident workgroups hide and a few hundred items do not -- 1920x1080: fix-up +64 -> +54 us, 1280x720 +52 -> +43; from
        // ~2.4 MPix on the scalar-load form's higher occupancy wins (3840x2160: +124 against +141 us).  profiles/r06/fix_apply_ab.txt
        static const char *env_lds = SRCNN_DEBUG_ENV("SRCNN_DEBUG_FIX_LDS");           // A/B knob: 0 / 1 = never / always
        const bool lds_weights = env_lds ? std::atoi(env_lds) == 1
                                         : (long)p.width * (p.row_end - p.row_begin) * fix_frames <= 2400000L;
        HIP_TRY(c, launch_fixup(f, c->n_cu, c->fix_strict, lds_weights, c->stream));
    }
    return SRCNN_OK;
}

}  // namespace host
}  // namespace srcnn

extern "C" {

/* ------------------------- device-resident entry points -------------------- */

int srcnn_conv99x11_dev(srcnn_ctx *c, const uint8_t *d_src, size_t src_stride, float *d_planes,
                        size_t plane_stride, size_t plane_pitch, int width, int height)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!c->has_l12) return fail(c, SRCNN_ERR_STATE, "layers 1-2 not loaded (srcnn_set_weights / srcnn_conv99x11)");
    if (bad_plane(d_src, src_stride, width, height) || bad_plane(d_planes, plane_stride, width, height) ||
        plane_pitch < plane_stride * (size_t)height || bad_pitch(plane_pitch))
        return fail(c, SRCNN_ERR_INVALID, "conv99x11_dev: bad plane geometry");
    if (c->mode == SRCNN_MODE_EXACT) {
        HIP_TRY(c, launch_conv99x11_exact(d_src, (long)src_stride, 0, d_planes, (long)plane_stride,
                                          (long)plane_pitch, 0, width, height, 1,
                                          static_cast<const float *>(c->wraw.p), c->stream));
        return SRCNN_OK;
    }
    StripParams p{};
    p.src = d_src;
    p.src_stride = (long)src_stride;
    p.planes_out = d_planes;
    p.pl_stride = (long)plane_stride;
    p.pl_pitch = (long)plane_pitch;
    p.width = width;
    p.height = height;
    p.row_begin = 0;
    p.row_end = height;
    return run_strip(c, MODE_L12, p, 1);
}

int srcnn_conv55_dev(srcnn_ctx *c, const float *d_planes, size_t plane_stride, size_t plane_pitch,
                     uint8_t *d_dst, size_t dst_stride, int width, int height, float *d_preclamp)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!c->has_l3) return fail(c, SRCNN_ERR_STATE, "layer 3 not loaded (srcnn_set_weights / srcnn_conv55)");
    if (bad_plane(d_planes, plane_stride, width, height) || bad_plane(d_dst, dst_stride, width, height) ||
        plane_pitch < plane_stride * (size_t)height || bad_pitch(plane_pitch))
        return fail(c, SRCNN_ERR_INVALID, "conv55_dev: bad plane geometry");
    if (c->mode == SRCNN_MODE_EXACT) {
        HIP_TRY(c, launch_conv55_exact(d_planes, (long)plane_stride, (long)plane_pitch, 0, d_dst, d_preclamp,
                                       (long)dst_stride, 0, width, height, 1,
                                       static_cast<const float *>(c->wraw.p) + 7329, c->b3, c->stream));
        return SRCNN_OK;
    }
    StripParams p{};
    p.planes_in = d_planes;
    p.pl_stride = (long)plane_stride;
    p.pl_pitch = (long)plane_pitch;
    p.dst = d_dst;
    p.pre = d_preclamp;
    p.dst_stride = (long)dst_stride;
    p.width = width;
    p.height = height;
    p.row_begin = 0;
    p.row_end = height;
    return run_strip(c, MODE_L3, p, 1);
}

int srcnn_forward_y_unfused_dev(srcnn_ctx *c, const uint8_t *d_src, size_t src_stride, size_t src_frame_pitch,
                                uint8_t *d_dst, size_t dst_stride, size_t dst_frame_pitch, int width,
                                int height, int n_frames, float *d_work)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!has_model(c)) return fail(c, SRCNN_ERR_STATE, "%s", kNoModel);
    if (bad_plane(d_src, src_stride, width, height) || bad_plane(d_dst, dst_stride, width, height) || !d_work ||
        n_frames <= 0)
        return fail(c, SRCNN_ERR_INVALID, "forward_y_unfused_dev: bad arguments");
    static const char *env_plpad = SRCNN_DEBUG_ENV("SRCNN_DEBUG_PLPAD");     // experiment: floats added to the plane pitch
    const long pitch = (long)width * height + (env_plpad ? std::atol(env_plpad) : 0);
    if (bad_pitch((size_t)pitch)) return fail(c, SRCNN_ERR_INVALID, "forward_y_unfused_dev: plane too large");
    if (c->mode == SRCNN_MODE_EXACT) {
        HIP_TRY(c, launch_conv99x11_exact(d_src, (long)src_stride, (long)src_frame_pitch, d_work, width, pitch,
                                          32 * pitch, width, height, n_frames,
                                          static_cast<const float *>(c->wraw.p), c->stream));
        HIP_TRY(c, launch_conv55_exact(d_work, width, pitch, 32 * pitch, d_dst, nullptr, (long)dst_stride,
                                       (long)dst_frame_pitch, width, height, n_frames,
                                       static_cast<const float *>(c->wraw.p) + 7329, c->b3, c->stream));
        return SRCNN_OK;
    }
    StripParams p{};
    p.src = d_src;
    p.src_stride = (long)src_stride;
    p.src_frame_pitch = (long)src_frame_pitch;
    p.planes_out = d_work;
    p.pl_stride = width;
    p.pl_pitch = pitch;
    p.pl_frame_pitch = 32 * pitch;
    p.width = width;
    p.height = height;
    p.row_begin = 0;
    p.row_end = height;
    if ((rc = run_strip(c, MODE_L12, p, n_frames))) return rc;
    StripParams q{};
    q.planes_in = d_work;
    q.pl_stride = width;
    q.pl_pitch = pitch;
    q.pl_frame_pitch = 32 * pitch;
    q.dst = d_dst;
    q.dst_stride = (long)dst_stride;
    q.dst_frame_pitch = (long)dst_frame_pitch;
    q.width = width;
    q.height = height;
    q.row_begin = 0;
    q.row_end = height;
    return run_strip(c, MODE_L3, q, n_frames);
}

int srcnn_forward_y_dev(srcnn_ctx *c, const uint8_t *d_src, size_t src_stride, size_t src_frame_pitch,
                        uint8_t *d_dst, size_t dst_stride, size_t dst_frame_pitch, int width, int height,
                        int n_frames, float *d_preclamp)
{
    BIND_KEEP(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!has_model(c)) return fail(c, SRCNN_ERR_STATE, "%s", kNoModel);
    if (bad_plane(d_src, src_stride, width, height) || bad_plane(d_dst, dst_stride, width, height) ||
        n_frames <= 0)
        return fail(c, SRCNN_ERR_INVALID, "forward_y_dev: bad arguments");
    // every output pixel reads a 13x13 input window that other workgroups may already have overwritten
    if (ranges_overlap(d_src, span_elems(src_stride, src_frame_pitch, width, height, n_frames), d_dst,
                       span_elems(dst_stride, dst_frame_pitch, width, height, n_frames)))
        return fail(c, SRCNN_ERR_INVALID, "forward_y_dev: src and dst overlap (the path cannot run in place)");
    // (a pre-clamp request in REFBYTES mode wants the REFERENCE's float too: the exact kernels deliver both)
    if (c->mode == SRCNN_MODE_EXACT || ((c->mode == SRCNN_MODE_REFBYTES || c->mode == SRCNN_MODE_REFBYTES16) && d_preclamp)) {
        if ((rc = flush_seams(c))) return rc;
        // frame by frame through ONE 32-plane workspace (128 B/pixel), whatever the batch size
        const long pitch = (long)width * height;
        if ((rc = reserve(c, c->planes, (size_t)32 * pitch * 4))) return rc;
        float *work = static_cast<float *>(c->planes.p);
        for (int f = 0; f < n_frames; ++f) {
            HIP_TRY(c, launch_conv99x11_exact(d_src + (size_t)f * src_frame_pitch, (long)src_stride, 0, work, width,
                                              pitch, 0, width, height, 1, static_cast<const float *>(c->wraw.p),
                                              c->stream));
            HIP_TRY(c, launch_conv55_exact(work, width, pitch, 0, d_dst + (size_t)f * dst_frame_pitch,
                                           d_preclamp ? d_preclamp + (size_t)f * dst_frame_pitch : nullptr,
                                           (long)dst_stride, 0, width, height, 1,
                                           static_cast<const float *>(c->wraw.p) + 7329, c->b3, c->stream));
        }
        return SRCNN_OK;
    }
    // the seam scratch of a launch grows with its frames: frames_per_launch() bounds it
    const int kMaxFrames = frames_per_launch(c, width, height, n_frames);
    // A batch that runs as one single-plane launch per frame defers its own seam launches: frame k's seam blocks ride behind
    // frame k + 1's work items, the last frame's are queued before the call returns (unless the caller asked for deferral).
    const bool caller_defers = c->defer_seams;
    if (n_frames > 1 && kMaxFrames == 1) c->defer_seams = true;
    struct Restore {
        srcnn_ctx *c;
        bool v;
        ~Restore() { c->defer_seams = v; }
    } restore{c, caller_defers};
    for (int f0 = 0; f0 < n_frames; f0 += kMaxFrames) {
        StripParams p{};
        p.src = d_src + (size_t)f0 * src_frame_pitch;
        p.src_stride = (long)src_stride;
        p.src_frame_pitch = (long)src_frame_pitch;
        p.dst = d_dst + (size_t)f0 * dst_frame_pitch;
        p.pre = d_preclamp ? d_preclamp + (size_t)f0 * dst_frame_pitch : nullptr;
        p.dst_stride = (long)dst_stride;
        p.dst_frame_pitch = (long)dst_frame_pitch;
        p.width = width;
        p.height = height;
        p.row_begin = 0;
        p.row_end = height;
        const bool refbytes = c->mode == SRCNN_MODE_REFBYTES || c->mode == SRCNN_MODE_REFBYTES16;      // (kMaxFrames is 1)
        // frames per fix-up: the work lists' 32-bit pixel codes (frame * height + y) * width + x must not wrap
        const int fix_batch = (int)std::max<unsigned long long>(
            1ull, std::min<unsigned long long>(FIX_BATCH_FRAMES, 0xffffffffull / ((unsigned long long)width * height)));
        const int batch0 = f0 - f0 % fix_batch;
        if ((rc = run_strip(c, MODE_FUSED, p, std::min(kMaxFrames, n_frames - f0), refbytes ? f0 - batch0 : 0,
                            refbytes ? std::min(fix_batch, n_frames - batch0) : 1, /*may_defer=*/true)))
            return rc;
    }
    if (!caller_defers && (rc = flush_seams(c))) return rc;
    return SRCNN_OK;
}

int srcnn_forward_y_rows_dev(srcnn_ctx *c, const uint8_t *d_src, size_t src_stride, int src_row0,
                             uint8_t *d_dst, size_t dst_stride, int dst_row0, int width, int height,
                             int row_begin, int row_end)
{
    BIND_KEEP(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!has_model(c)) return fail(c, SRCNN_ERR_STATE, "%s", kNoModel);
    if (bad_plane(d_src, src_stride, width, height) || bad_plane(d_dst, dst_stride, width, height) ||
        row_begin < 0 || row_end > height || row_begin >= row_end ||
        src_row0 > std::max(0, row_begin - 6) || dst_row0 > row_begin || src_row0 < 0 || dst_row0 < 0)
        return fail(c, SRCNN_ERR_INVALID, "forward_y_rows_dev: bad arguments");
    if (c->mode == SRCNN_MODE_EXACT) return fail(c, SRCNN_ERR_STATE, "row stripes are MFMA-mode only");
    StripParams p{};
    p.src = d_src;
    p.src_stride = (long)src_stride;
    p.src_row0 = src_row0;
    p.dst = d_dst;
    p.dst_stride = (long)dst_stride;
    p.dst_row0 = dst_row0;
    p.width = width;
    p.height = height;
    p.row_begin = row_begin;
    p.row_end = row_end;
    return run_strip(c, MODE_FUSED, p, 1, 0, 1, /*may_defer=*/true);
}

int srcnn_forward_y_rows_halo_dev(srcnn_ctx *c, const uint8_t *d_src, size_t src_stride, int src_row0, int src_rows,
                                  const uint8_t *d_halo_top, const uint8_t *d_halo_bot, size_t halo_stride,
                                  uint8_t *d_dst, size_t dst_stride, int dst_row0, int width, int height,
                                  int row_begin, int row_end)
{
    BIND_KEEP(c);
    if (!has_model(c)) return fail(c, SRCNN_ERR_STATE, "%s", kNoModel);
    const int src_row1 = src_row0 + src_rows;
    if (bad_plane(d_src, src_stride, width, height) || bad_plane(d_dst, dst_stride, width, height) || src_rows <= 0 ||
        row_begin < 0 || row_end > height || row_begin >= row_end || src_row0 < 0 || src_row1 > height ||
        dst_row0 > row_begin || dst_row0 < 0 || ((d_halo_top || d_halo_bot) && halo_stride < (size_t)width) ||
        halo_stride >= ((size_t)1 << 30))
        return fail(c, SRCNN_ERR_INVALID, "forward_y_rows_halo_dev: bad arguments");
    // the 13x13 receptive field of rows [row_begin, row_end) must lie in top | src | bot
    const int need0 = std::max(0, row_begin - kHaloRows), need1 = std::min(height, row_end + kHaloRows);
    if ((need0 < src_row0 && (!d_halo_top || src_row0 < kHaloRows || need0 < src_row0 - kHaloRows)) ||
        (need1 > src_row1 && (!d_halo_bot || need1 > src_row1 + kHaloRows)))
        return fail(c, SRCNN_ERR_INVALID, "forward_y_rows_halo_dev: rows [%d,%d) need input rows [%d,%d); src holds [%d,%d) and "
                                          "the halo buffers 6 rows either side", row_begin, row_end, need0, need1, src_row0, src_row1);
    if (c->mode != SRCNN_MODE_MFMA && c->mode != SRCNN_MODE_REFBYTES)
        return fail(c, SRCNN_ERR_STATE, "separate halo buffers are read by the float32 MFMA kernel only (SRCNN_MODE_MFMA / REFBYTES)");
    {   // like srcnn_forward_y_dev: the path cannot run in place -- the rows written must overlap none of the rows read
        const uint8_t *out0 = d_dst + (size_t)(row_begin - dst_row0) * dst_stride;
        const size_t out_bytes = (size_t)(row_end - row_begin - 1) * dst_stride + (size_t)width;
        const size_t halo_bytes = (size_t)(kHaloRows - 1) * halo_stride + (size_t)width;
        if (ranges_overlap(out0, out_bytes, d_src, (size_t)(src_rows - 1) * src_stride + (size_t)width) ||
            (d_halo_top && ranges_overlap(out0, out_bytes, d_halo_top, halo_bytes)) ||
            (d_halo_bot && ranges_overlap(out0, out_bytes, d_halo_bot, halo_bytes)))
            return fail(c, SRCNN_ERR_INVALID, "forward_y_rows_halo_dev: the output rows overlap the input (src or a halo buffer)");
    }
    StripParams p{};
    p.src = d_src;
    p.src_stride = (long)src_stride;
    p.src_row0 = src_row0;
    p.src_row1 = src_row1;
    // a side the launch reads nothing from keeps a null pointer: with both null this is srcnn_forward_y_rows_dev
    p.src_top = need0 < src_row0 ? d_halo_top : nullptr;
    p.src_bot = need1 > src_row1 ? d_halo_bot : nullptr;
    p.halo_stride = (long)halo_stride;
    p.dst = d_dst;
    p.dst_stride = (long)dst_stride;
    p.dst_row0 = dst_row0;
    p.width = width;
    p.height = height;
    p.row_begin = row_begin;
    p.row_end = row_end;
    return run_strip(c, MODE_FUSED, p, 1, 0, 1, /*may_defer=*/true);
}


}  // extern "C"
