// pt_launch.cpp -- tree -> launches (host side of libptamd.so; shared declarations: pt_context.hpp).
//   * where the traversal reads its nodes from: whole tree in LDS, treelet, 4-wide nodes from global memory; stack sizing
//   * kernel parameters (RenderParams), event timing
//   * the launch policy: kernel instance, schedule, samples per work item, by tiles per resident wave (DESIGN.md sections 5, 6)
//   * the wavefront variant's chains; the work counter's recovery after a launch that did not run to its end
//   * pt_generate_rays, pt_trace_rays, pt_render, pt_sync, pt_debug_launch_plan
#include "pt_context.hpp"

namespace ptamd {

// rows owned by `rank`: r with (r / rb) % world == rank
int32_t count_local_rows(int32_t H, int32_t rank, int32_t world, int32_t rb) {
    int32_t n = 0;
    for (int32_t r = 0; r < H; ++r)
        if ((r / rb) % world == rank) ++n;
    return n;
}

// 4-wide nodes: a visit finds at most `pending` entries above the sentinel and stores its three other children above
// the top, kept or not (Trav::wide_step)
int wide_stack_entries(int pending) { return ((pending + 4) + 1) & ~1; }
int stack_entries_for(int interior_depth) { return std::min(kStackEntries, ((interior_depth + 2) + 1) & ~1); }


// Does the whole tree fit next to two 512-thread workgroups per CU (kNodesLds: 16-bit references)?
bool whole_tree_fits_lds(size_t n_nodes, size_t n_tris, int interior_depth, int n_flat) {
    const bool s16 = n_nodes <= 32767 && n_tris <= 4096;
    const size_t block = (size_t)kLdsBlockBase;      // (the 768-thread k_render instance is taken only where it fits too)
    return s16 && (size_t)kLdsNodeBytes * n_nodes + 16 + (size_t)stack_entries_for(interior_depth) * 2 * block + (size_t)n_flat * 100 + 64 <= kLdsPerCu / 2;   // (+ flat list: packet + box + group mask per triangle)
}

// The listed triangles whose bounding boxes coincide -- the two halves of an axis-aligned wall -- are culled with ONE
// box test (Trav::flat_pass): group them by the vertex extremes the kernel pads into that box.
static void group_flat_boxes(pt_context* ctx) {
    ctx->n_fbox = 0;
    float ext[32][6];
    for (int k = 0; k < ctx->n_flat; ++k) {
        const float* v = ctx->packets[(size_t)k].v;          // r1, r2, r3
        float e[6];
        for (int a = 0; a < 3; ++a) {
            e[a] = std::min(std::min(v[a], v[3 + a]), v[6 + a]);
            e[3 + a] = std::max(std::max(v[a], v[3 + a]), v[6 + a]);
        }
        int b = 0;
        for (; b < ctx->n_fbox; ++b)
            if (std::memcmp(ext[b], e, sizeof e) == 0) break;
        if (b == ctx->n_fbox) {
            std::memcpy(ext[b], e, sizeof e);
            ctx->fbox_rep[b] = (uint8_t)k;
            ctx->fbox_mask[b] = 0;
            ctx->n_fbox++;
        }
        ctx->fbox_mask[b] |= 1u << k;
    }
}

// d_bvh2: the tree as it is in ctx->nodes, already in device memory (a device-built tree) -- the 4-wide collapse then runs
// there too (pt_widedev.hip: the same nodes) and *wide_on_device reports that ctx->d_nodes4 is in place.
int plan_node_placement(pt_context* ctx, const float4* d_bvh2, bool* wide_on_device) {
    PhaseClock clk("node placement");
    if (wide_on_device) *wide_on_device = false;
    ctx->treelet_nodes = 0;
    group_flat_boxes(ctx);
    ctx->interior_depth = deepest_interior_node(ctx->nodes);
    clk.lap("list boxes + interior depth");
    if (ctx->interior_depth + 2 > kStackEntries) return fail(ctx, PT_ESCENE, "internal: BVH deeper than the traversal stack");
    const bool fits = whole_tree_fits_lds(ctx->nodes.size(), ctx->orig.size(), ctx->interior_depth, ctx->n_flat);
    if (ctx->treelet != 0 && !fits) ctx->treelet_nodes = reindex_treelet(ctx->nodes, ctx->interior_depth, ctx->treelet);
    // 4-wide nodes for trees read from global memory -- unless their worst-case stack would not leave room for four
    // 256-thread workgroups per CU (then the BVH2 path stays)
    ctx->nodes4.clear();
    ctx->wide_pending = 0;
    if (ctx->wide_nodes == 2 || (ctx->wide_nodes == 1 && !fits && ctx->treelet_nodes == 0)) {
        if (d_bvh2 && ctx->treelet_nodes == 0 && ctx->wide_on_device != 0) {
            float4* d4 = nullptr;
            int n4 = 0, pending = 0;
            bool failed = false;
            PT_HIP(ctx, wide_device_build(d_bvh2, (int)ctx->nodes.size(), ctx->stream, &d4, &n4, &pending, &failed));
            if (!failed) {
                if (ctx->d_nodes4) (void)hipFree(ctx->d_nodes4);
                ctx->d_nodes4 = d4;
                ctx->nodes4.resize((size_t)n4);          // host copy for the debug getter and the stack sizing
                ctx->wide_pending = pending;
                PT_HIP(ctx, hipMemcpy(ctx->nodes4.data(), d4, sizeof(Node4q) * (size_t)n4, hipMemcpyDeviceToHost));
                if (wide_on_device) *wide_on_device = true;
            }
            clk.lap("4-wide nodes (device)");
            return PT_OK;
        }
        const unsigned hw = std::thread::hardware_concurrency();
        const int threads = ctx->build_threads > 0 ? ctx->build_threads : (int)std::min(16u, std::max(1u, hw));
        if (!build_wide_nodes(ctx->nodes, &ctx->nodes4, &ctx->wide_pending, threads)) ctx->nodes4.clear();
        clk.lap("4-wide nodes (host)");
    }
    return PT_OK;
}

int alloc_stack_overflow(pt_context* ctx) {
    if (ctx->d_stack_ovf) { PT_HIP(ctx, hipFree(ctx->d_stack_ovf)); ctx->d_stack_ovf = nullptr; }
    ctx->stack_ovf_lanes = 0;
    const int extra = ctx->nodes4.empty() ? 0 : wide_stack_entries(ctx->wide_pending) - ctx->wide_lds_entries;
    if (extra <= 0) return PT_OK;
    const size_t n_tiles = (size_t)((ctx->W + 7) / 8) * (size_t)((ctx->local_rows + 7) / 8);
    // every grid that can be in flight at once: k_render (one thread per pixel at most), or kWfDefaultChains concurrent wf_intersect
    // launches of the wavefront variant (2 cost classes x 6 x 256 threads per CU each, render_wavefront)
    const size_t lanes = 256 * std::max<size_t>((size_t)ctx->cu_count * 12 * kWfDefaultChains, (n_tiles + 3) / 4);
    PT_HIP(ctx, hipMalloc((void**)&ctx->d_stack_ovf, lanes * (size_t)extra * sizeof(uint32_t)));
    ctx->stack_ovf_lanes = lanes;
    return PT_OK;
}

int seed_upload(pt_context* ctx, const int32_t* global_seeds) {
    std::vector<int32_t> local((size_t)ctx->npix);
    for (int32_t lr = 0; lr < ctx->local_rows; ++lr) {
        const int32_t gr = global_row(ctx, lr);
        std::memcpy(&local[(size_t)lr * ctx->W], &global_seeds[(size_t)gr * ctx->W], sizeof(int32_t) * (size_t)ctx->W);
    }
    if (ctx->npix) PT_HIP(ctx, hipMemcpyAsync(ctx->d_rnds, local.data(), sizeof(int32_t) * local.size(), hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

void fill_params(const pt_context* ctx, const pt_camera* cam, RenderParams* p) {
    std::memset(p, 0, sizeof *p);
    p->nodes = ctx->d_nodes;
    p->tris = ctx->d_tris;
    p->meta = ctx->d_meta;
    p->mats = ctx->d_mats;
    p->rnds = ctx->d_rnds;
    p->colors = ctx->d_colors;
    p->rays = ctx->d_rays;
    p->stats = ctx->d_stats;
    p->cam = *cam;
    p->width = ctx->W;
    p->height = ctx->H;
    p->local_rows = ctx->local_rows;
    p->rank = ctx->rank;
    p->world = ctx->world;
    p->rows_per_block = ctx->rows_per_block;
    p->n_nodes = (int32_t)ctx->nodes.size();
    p->n_tris = (int32_t)ctx->orig.size();
    p->n_flat = ctx->n_flat;
    p->n_fbox = ctx->n_fbox;
    std::memcpy(p->fbox_rep, ctx->fbox_rep, sizeof p->fbox_rep);
    std::memcpy(p->fbox_mask, ctx->fbox_mask, sizeof p->fbox_mask);
    p->stack_entries = stack_entries_for(ctx->interior_depth);
    p->stack_ovf = nullptr;
    p->stack_ovf_lanes = 0;
    // where the traversal reads nodes from: the whole tree staged in LDS, its re-indexed top, or L1/L2 only
    p->node_mode = kNodesGlobal;
    p->treelet_nodes = 0;
    if (ctx->lds_scene) {
        if (whole_tree_fits_lds(ctx->nodes.size(), ctx->orig.size(), ctx->interior_depth, ctx->n_flat)) {
            p->node_mode = kNodesLds;
        } else if (ctx->treelet_nodes > 0) {
            p->node_mode = kNodesTreelet;
            p->treelet_nodes = ctx->treelet_nodes;
        }
    }
    if (!ctx->nodes4.empty() && (p->node_mode == kNodesGlobal || ctx->wide_nodes == 2)) {
        p->node_mode = kNodesWide;
        p->treelet_nodes = 0;
        p->nodes = ctx->d_nodes4;
        p->stack_entries = std::min(ctx->wide_lds_entries, wide_stack_entries(ctx->wide_pending));
        p->stack_ovf = ctx->d_stack_ovf;
        p->stack_ovf_lanes = (int32_t)ctx->stack_ovf_lanes;
    }
    p->tile_counter = nullptr;
    p->poll_ticks = (uint32_t)std::min<int64_t>((int64_t)ctx->poll_timeout_ms * 100000, 0xffffffffll);      // s_memrealtime: 100 MHz
    p->debug_stall_tile = ctx->debug_stall_tile;
    p->chunk_spp = 0;
    p->n_taper = 0;
    p->tile_done = nullptr;
    p->n_tiles = ((ctx->W + 7) / 8) * ((ctx->local_rows + 7) / 8);
    // measured with the big-triangle list in place (profiles/r02/i_*): Cornell box in LDS 8 / 16 / 24 / 32 -> 1669 / 1690 /
    // 1680 / 1669 Msamples/s (lockstep 1671); MESH-100k 24 / 32 / 48 -> 600 / 599 / 595 (lockstep 571); MESH-1M 211 / 210 / 205 (191)
    // (re-swept at the end of the round: Cornell box 8 / 12 / 16 / 20 / 24 / 32 -> 1957 / 2033 / 2066 / 2074 / 2078 / 2070)
    // (round 4, with the phase switching below: tree in LDS 12 / 16 / 20 / 24 / 32 -> 2637 / 2638 / 2621 / 2607 / 2554; from global
    // memory the rate is flat from 16 to 24: profiles/r04/)
    p->suspend_lanes = ctx->suspend_lanes >= 0 ? ctx->suspend_lanes : (p->node_mode == kNodesLds ? 16 : 24);
    // a phase of a while-while round ends early when at most this many lanes are still in it and some lane has left it
    // (Trav::round).  1080p, node_min / leaf_min (profiles/r04/c_*): tree in LDS (Cornell box) 0/0 2436, 3/4 2589, 4/8 2590, 8/4 2506
    // Msamples/s; 4-wide nodes from global memory 0/0 839 | 294, 4/4 1012 | 367, 6/4 1028 | 373, 8/8 1031 | 371 (MESH-100k | MESH-1M)
    p->node_min_lanes = ctx->node_min_lanes >= 0 ? ctx->node_min_lanes : (p->node_mode == kNodesLds ? 3 : 6);
    p->leaf_min_lanes = ctx->leaf_min_lanes >= 0 ? ctx->leaf_min_lanes : 4;
    p->migrate_lanes = ctx->migrate_lanes > 0 ? ctx->migrate_lanes : 1;      // (1 / 9 / 20: MESH-100k 1,192 / 1,112 / 1,113 -- waiting for company costs more than moving alone)
}

int check_ready(pt_context* ctx, const pt_camera* cam) {
    if (!cam) return fail(ctx, PT_EINVAL, "camera is NULL");
    if (!ctx->tris_uploaded) return fail(ctx, PT_EINVAL, "pt_upload_triangles has not been called");
    if (!ctx->mats_uploaded) return fail(ctx, PT_EINVAL, "pt_upload_materials has not been called");
    if ((int32_t)cam->XM != ctx->W || (int32_t)cam->YM != ctx->H)
        return fail(ctx, PT_EINVAL, "camera XM/YM do not match the context's frame size");
    return PT_OK;
}

int time_begin(pt_context* ctx, EventPair** ep) {
    *ep = nullptr;
    if (!ctx->timing) return PT_OK;
    if (ctx->events_used == ctx->events.size()) {
        EventPair e;
        PT_HIP(ctx, hipEventCreate(&e.a));
        PT_HIP(ctx, hipEventCreate(&e.b));
        ctx->events.push_back(e);
    }
    *ep = &ctx->events[ctx->events_used++];
    PT_HIP(ctx, hipEventRecord((*ep)->a, ctx->stream));
    return PT_OK;
}
int time_end(pt_context* ctx, EventPair* ep) {
    ctx->kernel_launches++;
    if (ep) PT_HIP(ctx, hipEventRecord(ep->b, ctx->stream));
    return PT_OK;
}
int time_collect(pt_context* ctx) {
    for (size_t i = 0; i < ctx->events_used; ++i) {
        float ms = 0.f;
        PT_HIP(ctx, hipEventSynchronize(ctx->events[i].b));
        PT_HIP(ctx, hipEventElapsedTime(&ms, ctx->events[i].a, ctx->events[i].b));
        ctx->kernel_ms_acc += ms;
    }
    ctx->events_used = 0;
    return PT_OK;
}


// Wait for the context's stream, then look at what the kernels left behind: a launch that lost a hand-over between chained passes
// (k_render) has written the tile into the work counter's error word.  That, like any failed HIP call on the way, becomes PT_EHIP --
// and the work counter is cleared before the next launch (a launch that did not run to its end does not leave words 0 / 1 at zero).
int sync_and_check(pt_context* ctx) {
    PT_HIP(ctx, hipSetDevice(ctx->device));
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ctx->counters_suspect = true;
        return fail(ctx, PT_EHIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
    if (!ctx->launched_since_check || !ctx->d_tile_counter) return PT_OK;
    ctx->launched_since_check = false;
    uint32_t words[kTileCounterWords] = {};
    PT_HIP(ctx, hipMemcpy(words, ctx->d_tile_counter, sizeof words, hipMemcpyDeviceToHost));
    if (words[kTileCounterError] != 0) {
        ctx->counters_suspect = true;
        PT_HIP(ctx, hipMemset(ctx->d_tile_counter, 0, sizeof words));
        ctx->counters_suspect = false;
        char msg[256];
        std::snprintf(msg, sizeof msg, "k_render: pass %u of tile %u waited more than %d ms for the tile's previous pass to be handed over (the "
                      "wave rendering it was lost); the launch was wound down, the frame is incomplete", words[kTileCounterError + 1],
                      words[kTileCounterError] - 1, ctx->poll_timeout_ms);
        return fail(ctx, PT_EHIP, msg);
    }
    return PT_OK;
}

}  // namespace ptamd

extern "C" {

// The work counter of the persistent launches (d_tile_counter) is zeroed once, at pt_create: every launch that runs to its end leaves
// words 0 / 1 at zero (the last wave out resets them).  INVARIANT: a launch that did NOT run to its end -- a failed launch call, a
// failed synchronize, a lost hand-over -- sets counters_suspect, and the next launch clears the counter first.
static int prepare_work_counter(pt_context* ctx) {
    if (ctx->counters_suspect) {
        PT_HIP(ctx, hipMemsetAsync(ctx->d_tile_counter, 0, sizeof(uint32_t) * kTileCounterWords, ctx->stream));
        ctx->counters_suspect = false;
    }
    ctx->launched_since_check = true;
    return PT_OK;
}

static int ptamd_resident_waves(const pt_context*, const LaunchConfig& lc) { return lc.persistent_blocks * (lc.block / 64); }

// Samples per (pass, tile) work item of a persistent launch, by tiles per resident wave (0: whole tiles).
// fewer tiles per resident wave (1080p over 2 / 4 / 8 GPUs: 4.0 / 2.0 / 1.0): suspend with passes of 16, lockstep with
// passes of 8, lockstep with whole tiles (profiles/r02/q_*: 95.7 % / 87.5 % / 61.8 % of the one-GPU rate per GPU)
static int auto_chunk_spp(int n_tiles, int resident_waves, int nsamples, int schedule, int node_mode) {
    // schedule 2 has no tail at the end of an item to amortise, so its items can be short where that re-balances the chip; what is
    // left is the end of the LAUNCH, which auto_chunk_taper() shortens (profiles/r04/r_*, t_*)
    if (schedule == 2) return node_mode == kNodesLds ? 32 : nsamples >= 32 ? 16 : nsamples >= 16 ? 8 : nsamples >= 8 ? 4 : 0;
    return n_tiles >= 5 * resident_waves ? (nsamples >= 256 ? 64 : 32)
         : n_tiles >= 3 * resident_waves ? 16
         : n_tiles > resident_waves + resident_waves / 4 ? 8 : 0;
}
// The shortest pass of a launch whose last chunk's worth of samples is cut in halves (64 samples in passes of 32 -> 32, 16, 8, 8): the
// work counter hands items out pass by pass, so the launch ends on short items.  Only where short items cost nothing extra, i.e. under
// schedule 2 (under schedule 1 every item ends on its slowest pixel: Cornell box 2,612 -> 2,557 / 2,584 / 2,605 with 2 / 4 / 8), and only
// on launches long enough to have an end worth shortening (16 samples per launch: -1 ... -8 %).  64 samples per launch, schedule 2:
// Cornell box 32 | 32,16,8,8 -> 2,568 | 2,684; MESH-100k 8 x 8 | 16 x 3, 8, 4, 4 -> 1,208 | 1,234; MESH-1M 460 | 468 (profiles/r04/t_*)
// Lockstep items have no slowest pixel to end on either (the lanes of a wave finish a sample together), and the launches that run
// lockstep with chained passes -- the tile sets of ranks of 2 / 4, two to four tiles per resident wave -- are the ones whose end
// weighs most: passes of 8 ... 8, 4, 2, 2 instead of 8 x 8: a rank of 2 4,223 -> 4,294, a rank of 4 6,733 -> 6,881 Msamples/s whole job.
static int auto_chunk_taper(int nsamples, int schedule, int node_mode) {
    if (schedule == 0) return nsamples >= 32 ? 2 : 0;
    if (schedule != 2) return 0;
    return node_mode == kNodesLds ? (nsamples >= 64 ? 8 : 0) : (nsamples >= 32 ? 4 : 0);
}

static void launch_cfg(pt_context* ctx, const RenderParams& p, LaunchConfig* lc) {
    // whole tree in LDS: two 768-thread workgroups per CU (six waves per SIMD) if their LDS fits, else two of 512
    // -- and if the launch has at least TWO tiles for each of their waves: with fewer (a 1080p frame over 4 / 8 GPUs: 8,160 /
    // 4,080 tiles for 6,144 waves) the 128-VGPR instance, whose waves run each tile faster, wins (round 4, with the phase
    // switching: a rank of 4 6,806 against 6,515 Msamples/s whole job, a rank of 8 9,211 against 8,524; a rank of 2 -- 16,080
    // tiles -- 4,228 / 4,246 either way; profiles/r04/i_*)
    const bool wide_fits = p.node_mode == kNodesLds && 2 * (traversal_lds_bytes(p, kLdsBlockWide) + 512) <= kLdsPerCu;
    const bool wide_block = wide_fits && ctx->lds_block != kLdsBlockBase &&
                            (ctx->lds_block == kLdsBlockWide || p.n_tiles >= 2 * ctx->cu_count * 2 * (kLdsBlockWide / 64));
    lc->block = traversal_block(p.node_mode, wide_block);
    lc->lds_bytes = traversal_lds_bytes(p, lc->block);
    lc->count_work = ctx->count_work != 0;
    // Restart + tail suspension wins when a wave works through many tiles (one GPU, 1080p: 7.9 per resident wave:
    // Cornell +2.4 %, mesh scenes +13-18 %); with few tiles per wave the end of a tile -- its slowest pixels finishing
    // their last samples alone -- is on the critical path and lockstep, whose lanes finish together, wins clearly
    // (1080p over 4 / 8 ranks: 87 % / 62 % strong-scaling efficiency against 73 % / 42 %; over 2 ranks suspend wins again:
    // 95.7 % against 88 %; profiles/r02/e_*, q_*).
    // resident workgroups at 4 waves per SIMD: 2 x 512 threads (whole tree in LDS), 1 x 1024 (treelet) per CU; nodes through
    // L1/L2 (256 threads): as many waves per SIMD -- 7, 6, 5 or 4 -- as the stacks in LDS leave room for
    lc->waves_per_simd = wide_block ? kLdsWpsWide : 4;
    if (p.node_mode == kNodesGlobal || p.node_mode == kNodesWide) {
        const int want = ctx->waves_per_simd > 0 ? ctx->waves_per_simd : 7;
        for (int w = std::min(want, 8); w > 4; --w)
            if ((size_t)w * lc->lds_bytes + 1024 <= kLdsPerCu) { lc->waves_per_simd = w; break; }
    }
    lc->persistent_blocks = ctx->cu_count * std::max(1, 256 * lc->waves_per_simd / lc->block);
    lc->cu_count = ctx->cu_count;
    // ... and with few samples per launch: a lane has no next sample to start while the others finish, and lanes that run
    // ahead give up the coherence of a tile's camera rays -- lockstep up to 4 samples per launch with the tree in LDS
    // (render(1): 1,810 against 1,492 Msamples/s), for one sample otherwise (profiles/r03/q_*)
    const bool few_samples = p.nsamples <= (p.node_mode == kNodesLds ? 4 : 1);
    // Round 4: the restart schedule is the one whose lanes also move on to the wave's next work item instead of waiting for the item's
    // slowest pixel (schedule 2, render_items_migrating): MESH-100k 1,099 -> 1,190, MESH-1M 423 -> 456 Msamples/s at 64 samples per
    // launch.  The Cornell box gains nothing from that by itself (its lanes finish together: 2,646 -> 2,624) but, like the meshes, from
    // the short last passes that only schedule 2 makes free (auto_chunk_taper: 2,613 -> 2,684, from 64 samples per launch; with 16
    // schedule 1 leads 2,392 to 2,323) (profiles/r04/r_*, t_*).
    const int restart = (ctx->persistent && (p.node_mode != kNodesLds || p.nsamples >= 64)) ? 2 : 1;
    lc->schedule = ctx->schedule >= 0 ? ctx->schedule : (!few_samples && p.n_tiles >= 3 * ptamd_resident_waves(ctx, *lc) ? restart : 0);
    ctx->last_lds_bytes = lc->lds_bytes;
    ctx->last_waves_per_simd = lc->waves_per_simd;
}

int pt_generate_rays(pt_context* ctx, const pt_camera* cam) {
    PT_NEED_DEVICE(ctx);
    int rc = check_ready(ctx, cam);
    if (rc != PT_OK) return rc;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    RenderParams p;
    fill_params(ctx, cam, &p);
    PT_HIP(ctx, launch_gen_ray(p, ctx->stream));
    return PT_OK;
}

int pt_trace_rays(pt_context* ctx, const pt_camera* cam, int32_t iterations, int32_t current_sample) {
    PT_NEED_DEVICE(ctx);
    int rc = check_ready(ctx, cam);
    if (rc != PT_OK) return rc;
    if (iterations < 0 || current_sample < 0) return fail(ctx, PT_EINVAL, "iterations/current_sample must be >= 0");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    RenderParams p;
    fill_params(ctx, cam, &p);
    p.iterations = iterations;
    p.first_sample = current_sample;
    p.nsamples = 1;
    LaunchConfig lc;
    launch_cfg(ctx, p, &lc);
    if (ctx->persistent) {      // the grid only fills the chip: a workgroup stages the tree once, not once per eight tiles
        if (int rc2 = prepare_work_counter(ctx)) return rc2;
        p.tile_counter = ctx->d_tile_counter;      // (zero: the previous launch's last wave reset it)
    }
    EventPair* ep;
    if ((rc = time_begin(ctx, &ep)) != PT_OK) return rc;
    {
        const hipError_t le = launch_trace_ray(p, lc, ctx->stream);
        if (le != hipSuccess) { ctx->counters_suspect = true; return fail(ctx, PT_EHIP, std::string("launch_trace_ray: ") + hipGetErrorString(le)); }
    }
    ctx->render_epoch++;
    rc = time_end(ctx, ep);
    if (rc != PT_OK) ctx->counters_suspect = true;
    return rc;
}

// The stream-compacted variant.  The local pixels are cut into `wf_streams` contiguous chains; every chain owns its ray streams,
// hit stream, class queues and counters and runs its passes -- wf_generate, then per bounce wf_intersect and wf_shade -- on a HIP
// stream of its own.  A pass is a chain of 17 dependent launches of ~100 us, and a persistent wf_intersect launch ends on its
// longest ray: ~60 us of the 133 us a 1080p launch took were the same at a quarter and at four times the rays
// (profiles/r04/).  With two chains the tail of one runs under the body of the other.
static int render_wavefront(pt_context* ctx, const RenderParams& rp, int32_t nsamples) {
    if (rp.iterations > kWfMaxBounces) return fail(ctx, PT_EINVAL, "wavefront variant supports at most 1023 iterations");
    const size_t np = (size_t)std::max<int64_t>(ctx->npix, 1);
    constexpr size_t kCounterWords = (size_t)kWfCounterStride * (kWfMaxBounces + 4);
    if (!ctx->d_wf_state) {
        PT_HIP(ctx, hipMalloc((void**)&ctx->d_wf_state, sizeof(float4) * 17 * np));
        PT_HIP(ctx, hipMalloc((void**)&ctx->d_wf_queues, sizeof(int32_t) * 3 * np));
        PT_HIP(ctx, hipMalloc((void**)&ctx->d_wf_counters, sizeof(uint32_t) * kCounterWords * kWfMaxChains));
    }
    if (ctx->npix == 0) return PT_OK;
    // chains of whole 8,192-pixel units, at least ~64k pixels each (a chain of a few thousand rays is all launch overhead)
    const int want = ctx->wf_streams > 0 ? ctx->wf_streams : kWfDefaultChains;
    int chains = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(want, kWfMaxChains), ctx->npix / 65536));
    // 4-wide traversal: the part of a lane's stack that lies in global memory is addressed by the lane's place in the GRID; chains
    // run concurrently, so each gets its own range of that buffer (the widest wf_intersect grid: 2 cost classes x 6 x 256 threads per CU)
    const size_t chain_lanes = (size_t)2 * 6 * 256 * (size_t)ctx->cu_count;
    if (rp.stack_ovf) chains = (int)std::max<size_t>(1, std::min<size_t>((size_t)chains, (size_t)rp.stack_ovf_lanes / chain_lanes));
    for (int c = 1; c < chains; ++c)
        if (!ctx->wf_stream[c]) {
            PT_HIP(ctx, hipStreamCreateWithFlags(&ctx->wf_stream[c], hipStreamNonBlocking));
            PT_HIP(ctx, hipEventCreateWithFlags(&ctx->wf_event[c], hipEventDisableTiming));
        }
    if (!ctx->wf_event[0]) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->wf_event[0], hipEventDisableTiming));
    WfParams w[kWfMaxChains];
    // contiguous chains of whole 8,192-pixel units (dealing the frame out in interleaved blocks of 8 rows, so that every chain gets
    // its share of the cheap and the expensive parts, measured the same on the Cornell box and 30 % slower on MESH-1M, whose rays
    // then cover the whole mesh in every chain: profiles/r04/f_*)
    const int64_t unit = 8192;
    const int64_t per = ((ctx->npix + chains - 1) / chains + unit - 1) / unit * unit;
    for (int c = 0; c < chains; ++c) {
        WfParams& wc = w[c];
        const int64_t p0 = std::min<int64_t>((int64_t)c * per, ctx->npix), count = std::min<int64_t>(p0 + per, ctx->npix) - p0;
        wc.rp = rp;
        if (rp.stack_ovf) wc.rp.stack_ovf = rp.stack_ovf + (size_t)c * chain_lanes;      // (stack_ovf_lanes stays the stride between entries)
        wc.sP = reinterpret_cast<float*>(ctx->d_wf_state);        // kWfFields x np x 12 B <= 4 x np x 16 B, indexed by local pixel
        for (int par = 0; par < 2; ++par)
            for (int k = 0; k < 2; ++k) {
                wc.rsA[par][k] = ctx->d_wf_state + (size_t)(4 + (par * 2 + k) * 2 + 0) * np + p0;      // a chain's streams: its slice of each array
                wc.rsB[par][k] = ctx->d_wf_state + (size_t)(4 + (par * 2 + k) * 2 + 1) * np + p0;
                wc.rsC[par][k] = ctx->d_wf_state + (size_t)(13 + par * 2 + k) * np + p0;
            }
        wc.hit[0] = reinterpret_cast<float2*>(ctx->d_wf_state + 12 * np) + p0;
        wc.hit[1] = reinterpret_cast<float2*>(ctx->d_wf_state + 12 * np) + np + p0;
        for (int k = 0; k < 3; ++k) wc.q_cls[k] = ctx->d_wf_queues + (size_t)k * np + p0;
        wc.counters = ctx->d_wf_counters + (size_t)c * kCounterWords;
        wc.npix = (int32_t)count;
        wc.pix0 = (int32_t)p0;
        wc.npix_all = (int32_t)ctx->npix;
        wc.n_cbox = ctx->cost_binning ? (int32_t)(ctx->cost_boxes.size() / 6) : 0;
        for (int b = 0; b < wc.n_cbox; ++b)
            for (int k = 0; k < 6; ++k) wc.cbox[b][k] = ctx->cost_boxes[(size_t)b * 6 + k];
    }
    // the other chains' streams start behind whatever the context's stream holds, and the context's stream ends behind them
    PT_HIP(ctx, hipEventRecord(ctx->wf_event[0], ctx->stream));
    for (int c = 1; c < chains; ++c) PT_HIP(ctx, hipStreamWaitEvent(ctx->wf_stream[c], ctx->wf_event[0], 0));
    for (int32_t k = 0; k < nsamples; ++k) {
        for (int c = 0; c < chains; ++c) {
            if (w[c].npix == 0) continue;
            hipStream_t st = c == 0 ? ctx->stream : ctx->wf_stream[c];
            w[c].sample = rp.first_sample + k;
            PT_HIP(ctx, hipMemsetAsync(w[c].counters, 0, sizeof(uint32_t) * kWfCounterStride, st));
            PT_HIP(ctx, launch_wf_generate(w[c], st));
        }
        for (int32_t b = 0; b < rp.iterations; ++b)
            for (int c = 0; c < chains; ++c) {
                if (w[c].npix == 0) continue;
                hipStream_t st = c == 0 ? ctx->stream : ctx->wf_stream[c];
                EventPair* ep = nullptr;
                int rc = PT_OK;
                if (c == 0 && (rc = time_begin(ctx, &ep)) != PT_OK) return rc;      // kernel_ms: wf_intersect of chain 0 (the others overlap it)
                PT_HIP(ctx, launch_wf_intersect(w[c], b, ctx->cu_count, st));
                if (c == 0 && (rc = time_end(ctx, ep)) != PT_OK) return rc;
                PT_HIP(ctx, launch_wf_shade(w[c], b, st));
            }
        if (ctx->timing && ctx->events_used >= 4096) {   // bound the event pool
            int rc = time_collect(ctx);
            if (rc != PT_OK) return rc;
        }
    }
    for (int c = 1; c < chains; ++c) {
        PT_HIP(ctx, hipEventRecord(ctx->wf_event[c], ctx->wf_stream[c]));
        PT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->wf_event[c], 0));
    }
    return PT_OK;
}

int pt_render(pt_context* ctx, const pt_camera* cam, int32_t iterations, int32_t nsamples) {
    PT_NEED_DEVICE(ctx);
    int rc = check_ready(ctx, cam);
    if (rc != PT_OK) return rc;
    if (iterations < 0 || nsamples < 0) return fail(ctx, PT_EINVAL, "iterations/nsamples must be >= 0");
    if (nsamples == 0) return PT_OK;
    PT_HIP(ctx, hipSetDevice(ctx->device));
    RenderParams p;
    fill_params(ctx, cam, &p);
    p.iterations = iterations;
    p.first_sample = ctx->current_sample;
    p.nsamples = nsamples;
    ctx->render_epoch++;
    if (ctx->variant == 1) {
        if ((rc = render_wavefront(ctx, p, nsamples)) != PT_OK) return rc;
        ctx->current_sample += nsamples;
        return PT_OK;
    }
    LaunchConfig lc;
    launch_cfg(ctx, p, &lc);
    if (ctx->persistent) {
        if (int rc2 = prepare_work_counter(ctx)) return rc2;
        p.tile_counter = ctx->d_tile_counter;      // (zero: the previous launch's last wave reset it)
        // automatic: chaining passes only pays when there are enough tiles to re-balance; with about one
        // tile per resident wave (a 1080p frame over 8 GPUs) the most expensive tile is the critical path
        // either way and the extra hand-offs cost 2-4 %.  With many tiles per wave (one GPU, 1080p: 7.9) the
        // launch runs the suspend schedule, whose items end with their slowest pixels finishing alone: long
        // passes amortise that tail, and half a launch's samples still re-balance the chip -- 64 spp per launch:
        // 8 / 16 / 32 / 64 -> 1689 / 1760 / 1787 / 1761 Msamples/s; 256 spp: 1706 / 1789 / 1843 / 1868
        // (profiles/r02/n_*; the lockstep kernel of round 1 peaked at 8).
        const int resident_waves = ptamd_resident_waves(ctx, lc);
        // (schedule 2 has no tail at the end of an item to amortise: short passes, which re-balance the chip -- 4 / 8 / 16 / 32 samples per
        // item at 64 per launch: MESH-100k 1,138 / 1,190 / 1,186 / 1,146, MESH-1M 443 / 456 / 456 / 442; at 16 per launch 4 and 8 lead)
        const int auto_chunk = auto_chunk_spp(p.n_tiles, resident_waves, nsamples, lc.schedule, p.node_mode);
        const int chunk = ctx->chunk_spp >= 0 ? ctx->chunk_spp : auto_chunk;
        // work items are numbered with an int on the device: passes x tiles + one failed fetch per resident wave
        // passes of chunk samples, the last of them cut in halves down to `taper` samples: the launch ends on short items instead of
        // on whole ones (the work counter hands items out pass by pass, so the short ones ARE the end of the launch)
        int n_pass = chunk > 0 ? (int)(((int64_t)nsamples + chunk - 1) / chunk) : 1;
        const int taper = ctx->chunk_taper >= 0 ? ctx->chunk_taper : auto_chunk_taper(nsamples, lc.schedule, p.node_mode);
        if (chunk > 0 && nsamples > chunk && taper > 0 && taper < chunk && nsamples < 65536) {
            std::vector<int> ends;
            int done = 0;
            while (nsamples - done > chunk) { done += chunk; ends.push_back(done); }
            for (int left = nsamples - done; left > 0;) {
                const int take = left >= 2 * taper ? left / 2 : left;
                done += take;
                left -= take;
                ends.push_back(done);
            }
            if ((int)ends.size() <= kMaxTaperPasses) {
                n_pass = p.n_taper = (int)ends.size();
                for (int k = 0; k < n_pass; ++k) p.taper_end[k] = (uint16_t)ends[k];
            }
        }
        const int64_t items = chunk > 0 ? (int64_t)n_pass * p.n_tiles : p.n_tiles;
        if (items + (int64_t)resident_waves + 64 >= ((int64_t)1 << 31))
            return fail(ctx, PT_EINVAL, "nsamples / chunk_spp x tiles does not fit the 31-bit work-item counter of one launch: render in several calls");
        if (chunk > 0 && nsamples > chunk && p.n_tiles > 0) {
            if (!ctx->d_tile_done) PT_HIP(ctx, hipMalloc((void**)&ctx->d_tile_done, sizeof(uint32_t) * (size_t)p.n_tiles));
            PT_HIP(ctx, hipMemsetAsync(ctx->d_tile_done, 0, sizeof(uint32_t) * (size_t)p.n_tiles, ctx->stream));
            p.tile_done = ctx->d_tile_done;
            p.chunk_spp = chunk;
        }
    }
    if (lc.count_work && p.n_tiles > 0) {      // per-tile cost of this launch (pt_debug_tile_cost)
        if (!ctx->d_tile_cost) PT_HIP(ctx, hipMalloc((void**)&ctx->d_tile_cost, sizeof(uint32_t) * (size_t)p.n_tiles));
        PT_HIP(ctx, hipMemsetAsync(ctx->d_tile_cost, 0, sizeof(uint32_t) * (size_t)p.n_tiles, ctx->stream));
        p.tile_cost = ctx->d_tile_cost;
    }
    EventPair* ep;
    if ((rc = time_begin(ctx, &ep)) != PT_OK) return rc;
    {
        const hipError_t le = launch_render_mega(p, lc, ctx->stream);
        if (le != hipSuccess) { ctx->counters_suspect = true; return fail(ctx, PT_EHIP, std::string("launch_render_mega: ") + hipGetErrorString(le)); }
    }
    if ((rc = time_end(ctx, ep)) != PT_OK) { ctx->counters_suspect = true; return rc; }
    ctx->current_sample += nsamples;  // main.cpp:686
    return PT_OK;
}

int pt_set_current_sample(pt_context* ctx, int32_t s) {
    if (!ctx || s < 0) return PT_EINVAL;
    ctx->current_sample = s;
    return PT_OK;
}
int pt_get_current_sample(const pt_context* ctx, int32_t* out) {
    if (!ctx || !out) return PT_EINVAL;
    *out = ctx->current_sample;
    return PT_OK;
}


int pt_sync(pt_context* ctx) {
    PT_NEED_DEVICE(ctx);
    return sync_and_check(ctx);
}

// counting launches (option count_work): per 8x8 tile of the local frame, the shader-clock cycles / 64 its wave(s) spent on it,
// summed over the work items of the last pt_render call -- the latency a launch with one tile per wave ends on
// What pt_render(nsamples) would launch on a device of `cu_count` compute units (0: the context's own): works on a host-only
// context too, so that the launch policy -- which kernel instance, which schedule, which pass length a rank of an N-GPU job gets
// -- can be pinned by CPU tests.  out[8] = { threads per workgroup, waves per SIMD, schedule (0 lockstep, 1 suspend), samples per
// (pass, tile) work item (0: whole tiles), resident waves, tiles, node mode, dynamic LDS bytes }.
int pt_debug_launch_plan(pt_context* ctx, int32_t nsamples, int32_t cu_count, int64_t out[8]) {
    if (!ctx || !out || nsamples < 1 || cu_count < 0) return PT_EINVAL;
    if (!ctx->tris_uploaded) return fail(ctx, PT_EINVAL, "upload_Triangles first");
    pt_camera cam;
    const float shift[3] = {0.f, 0.f, 0.f};
    pt_camera_init(&cam, 60.0f, 0.0f, 0.0f, shift, ctx->W, ctx->H);
    RenderParams p;
    fill_params(ctx, &cam, &p);
    p.nsamples = nsamples;
    const int saved_cu = ctx->cu_count;
    const size_t saved_lds = ctx->last_lds_bytes;
    const int saved_wps = ctx->last_waves_per_simd;
    if (cu_count > 0) ctx->cu_count = cu_count;
    LaunchConfig lc;
    launch_cfg(ctx, p, &lc);
    const int resident = ptamd_resident_waves(ctx, lc);
    ctx->cu_count = saved_cu;
    ctx->last_lds_bytes = saved_lds;
    ctx->last_waves_per_simd = saved_wps;
    const int chunk = ctx->chunk_spp >= 0 ? ctx->chunk_spp : auto_chunk_spp(p.n_tiles, resident, nsamples, lc.schedule, p.node_mode);
    out[0] = lc.block; out[1] = lc.waves_per_simd; out[2] = lc.schedule; out[3] = (chunk > 0 && nsamples > chunk) ? chunk : 0;
    out[4] = resident; out[5] = p.n_tiles; out[6] = p.node_mode; out[7] = (int64_t)lc.lds_bytes;
    return PT_OK;
}


}  // extern "C"
