// pt_kernels.hip -- the render kernels of libptamd.so (gfx950).
//
// The hot path of the reference (prog.cl:292-389: gen_ray + trace_ray) as one persistent
// "render" kernel: every lane owns one pixel and runs ALL requested samples of it back to
// back with the whole path state in registers (the per-pixel LCG stream is sequential across
// samples, prog.cl:72-77 + main.cpp:382, so a pixel's samples can never run concurrently
// anyway).  A lane whose path ends regenerates its next camera ray in the same loop, so the
// wave stays converged on "traverse -> shade" instead of idling until the longest path of
// the wave is done.  The launch is persistent: waves pull (pass, tile) work items from a global
// counter (k_render); rnds/colors travel through HBM once per pass of 32-64 samples (~1.8 B/sample).
//
// Traversal (pt_device.hpp): own BVH2 (64-B nodes holding both child boxes, 48-B triangle packets) or its
// 4-wide quantised collapse; the biggest triangles (walls, floor ...) are kept out of the tree and tested
// first from an LDS copy; near child first, far children on a per-lane stack in LDS ([entry][lane]).
//   * whole tree fits LDS (Cornell box: 935 nodes x 56 B): every workgroup stages it, re-laid out so that the planes a ray
//     needs are picked by address; two 768-thread workgroups per CU = six waves per SIMD at 80 VGPRs (two of 512 at
//     128 VGPRs where the stacks leave no room, or where the launch has fewer tiles than that many waves);
//   * larger trees: nodes (4-wide by default) and packets through L1 / L2, 256-thread workgroups, up to seven waves
//     per SIMD at 72 VGPRs (LEAN: nothing recomputable is carried across a traversal, double-precision constants sit
//     in scalar registers); optionally the top of the tree in LDS behind a generic pointer (`treelet`).
// Every instance keeps its wave-uniform loop state in scalar registers; `make check-isa` verifies that in the ISA.
#include "pt_device.hpp"


#include <algorithm>

namespace ptamd {

// ---------------------------------------------------------------------------- kernels
// gen_ray, prog.cl:384-389
__global__ void __launch_bounds__(256) k_gen_ray(RenderParams p) {
    const PixelId px = pixel_of_thread(p);
    if (px.li < 0) return;
    int seed = p.rnds[px.li];
    const float rnd1 = lcg_rand(seed), rnd2 = lcg_rand(seed);
    f3 P, D;
    camera_get_ray(px.gid, p.cam, rnd1, rnd2, &P, &D);
    p.rnds[px.li] = seed;
    float4* r = reinterpret_cast<float4*>(&p.rays[px.li]);
    r[0] = make_float4(P.x, P.y, P.z, 0.0f);
    r[1] = make_float4(D.x, D.y, D.z, 0.0f);
}

// All samples [s_begin, s_end) of ONE pixel, path state in registers: nsamples x (gen_ray + trace_ray).
// SPLIT: trace_ray alone (prog.cl:292-381) -- the ray comes from, and is left in, the rays buffer.
//
// Two schedules of the same per-pixel computation (what a wave executes together is the only difference;
// tools/sim/policy_sim.py replays both on recorded traversal traces and predicts their instruction counts
// within a few percent of the SQ counters, profiles/r02/):
//
//  kSchedLockstep  explicit sample loop around a bounce loop.  All lanes of a wave start a sample together and
//                  a lane whose path ended waits for the wave's longest path.  Idle lanes cost nothing in a
//                  VALU-issue-bound kernel, while lanes that march in step trace rays of the same generation:
//                  the camera rays of a tile are coherent (their traversal executes the node body 10 times per
//                  wave instead of 54), first-bounce rays still start next to each other (35).  Best schedule
//                  for the Cornell box (whole tree in LDS): 7-11 % fewer instructions than "restart".
//
//  kSchedSuspend   one flat segment loop: a lane whose path ended starts its next sample at once ("restart"),
//                  and the traversal is resumable (tail suspension).  The lanes of a wave need very different
//                  numbers of node visits (median 6, 99th percentile 33: over half of the node-body executions
//                  of a plain while-while loop run for <= 4 lanes), so the wave leaves the traversal as soon as
//                  at most p.suspend_lanes lanes are still at it (and at least one has finished): the finished
//                  lanes shade and start their next segment, the stragglers keep {best_t, best, tos, cur} and
//                  resume in the next trip, where their remaining visits overlap with everybody's new rays.
//                  Costs more shading executions (each for fewer lanes): a loss when VALU-bound, +13-14 % on
//                  the latency-bound mesh scenes where every wave-level step saved is a memory round trip saved.
//
//  kSchedMigrate   kSchedSuspend without the END of a work item.  A wave's item ends on its slowest pixel: the lanes' samples
//                  are paths of 1 .. iterations segments, so after 32 samples the busiest lane of 64 has ~12 % more segments
//                  behind it than the average one, and the last of them run for a handful of lanes (at 3-4 times the cost per
//                  instruction, tools/micro/exec_ops.hip).  Here a lane whose pixel has had its samples writes it out and MOVES ON
//                  to its pixel of the wave's next work item, which the wave fetches as soon as the first lane asks for it, while
//                  the others finish: a wave has up to two items in hand, and only the end of the launch is a tail.  (render_items_migrating)
enum : int { kSchedLockstep = 0, kSchedSuspend = 1, kSchedMigrate = 2 };


// LEAN (the instances for 6 / 7 waves per SIMD: 80 / 72 VGPRs hold the traversal and little else): nothing that can be
// recomputed or fetched is carried across a traversal -- the running mean is folded into colors[] at the end of every
// sample, as prog.cl:379 does, instead of riding in three registers for the whole work item, and the pixel's float
// coordinates are rebuilt from its id at every sample start.  (The allocator spilled seven dwords around every
// traversal of the 72-VGPR instance: 16 GB written and ~38 GB re-read per 42-ms launch, profiles/r03/a_*.)
// prog.cl:379 on the frame buffer itself (LEAN): sample 0 starts from black (prog.cl:312-314)
PT_DEV void fold_sample(const RenderParams& p, int li, f3 color, int s) {
    f3 acc = mk(0.0f, 0.0f, 0.0f);
    if (s != 0) {
        const float4 c = p.colors[li];
        acc = mk(c.x, c.y, c.z);
    }
    acc = running_mean(acc, color, s);
    p.colors[li] = make_float4(acc.x, acc.y, acc.z, 0.0f);
}

template <bool SPLIT, int MODE, bool COUNT, bool LEAN, bool SK>
PT_DEV void render_pixel_lockstep(const RenderParams& p, const SceneView& sv, const LaneStack<typename StackOf<MODE>::type> stk, const PixelId px,
                                  int s_begin, int s_end, unsigned* segs, WorkCount* wc) {
    int seed = p.rnds[px.li];
    f3 acc = mk(0.0f, 0.0f, 0.0f);
    if (!LEAN && s_begin != 0) {               // prog.cl:312-314: sample 0 starts from black
        const float4 c = p.colors[px.li];
        acc = mk(c.x, c.y, c.z);
    }
    const int camX = (int)p.cam.XM;
    const int gx = px.gid % camX, gy = px.gid / camX;                                // prog.cl:84-85
    const unsigned pxy = (unsigned)gx | ((unsigned)gy << 16);                        // LEAN: one register (frames below 65,536 x 65,536: pt_create)
    const float pix_x0 = (float)gx, pix_y0 = (float)gy;
    f3 rP = mk(0.f, 0.f, 0.f), rD = mk(0.f, 0.f, 1.f);
    PathRegs st;
    st.reset();
    for (int s = s_begin; s < s_end; ++s) {    // the same trip count on every lane of the wave
        const float pix_x = LEAN ? (float)(pxy & 0xffffu) : pix_x0, pix_y = LEAN ? (float)(pxy >> 16) : pix_y0;
        st.reset();                            // prog.cl:307-316
        bool inside = false;
        if (SPLIT) {
            const float4* r = reinterpret_cast<const float4*>(&p.rays[px.li]);
            const float4 a = r[0], b = r[1];
            rP = mk(a.x, a.y, a.z);
            rD = mk(b.x, b.y, b.z);
        } else {
            const float rnd1 = lcg_rand(seed), rnd2 = lcg_rand(seed);
            camera_get_ray_xy(pix_x, pix_y, p.cam, rnd1, rnd2, &rP, &rD);
        }
        for (int bounce = 0; bounce < p.iterations; ++bounce) {     // a lane that leaves early waits for the others
            if (COUNT && first_active_lane()) wc->wtrips++;
            float t;
            const int ti = closest_hit<MODE, COUNT>(sv, rP, rD, stk, &t, wc);
            ++*segs;
            if (ti < 0) break;                                       // black environment, prog.cl:367-376
            if (COUNT) { if (first_active_lane()) wc->wshade++; count_low(wc, 3); }
            shade_hit<SK>(rP, rD, st, seed, inside, p, sv.tris, sv.meta, ti, t);
        }
        if (LEAN) fold_sample(p, px.li, st.C(), s);
        else acc = running_mean(acc, st.C(), s);
    }
    if (!LEAN) p.colors[px.li] = make_float4(acc.x, acc.y, acc.z, 0.0f);
    p.rnds[px.li] = seed;
    if (SPLIT) {
        float4* r = reinterpret_cast<float4*>(&p.rays[px.li]);
        r[0] = make_float4(rP.x, rP.y, rP.z, 0.0f);
        r[1] = make_float4(rD.x, rD.y, rD.z, 0.0f);
    }
}

template <bool SPLIT, int MODE, bool COUNT, bool LEAN, bool SK>
PT_DEV void render_pixel_suspend(const RenderParams& p, const SceneView& sv, const LaneStack<typename StackOf<MODE>::type> stk, const PixelId px,
                                 int s_begin, int s_end, unsigned* segs, WorkCount* wc) {
    f3 rP = mk(0.f, 0.f, 0.f), rD = mk(0.f, 0.f, 1.f);
    PathRegs st;
    st.reset();
    bool inside = false;
    int seed = p.rnds[px.li];
    f3 acc = mk(0.0f, 0.0f, 0.0f);
    if (!LEAN && s_begin != 0) {               // prog.cl:312-314: sample 0 starts from black
        const float4 c = p.colors[px.li];
        acc = mk(c.x, c.y, c.z);
    }
    int s = s_begin;
    int bounce = 0;
    bool fresh = true;
    bool traversing = false;                   // this lane holds a suspended traversal
    Trav<MODE> tr;
    tr.setup(rP, rD);
    tr.restart(stk);
    tr.idle();
    const int camX = (int)p.cam.XM;
    const int gx = px.gid % camX, gy = px.gid / camX;                                // prog.cl:84-85
    const unsigned pxy = (unsigned)gx | ((unsigned)gy << 16);                        // LEAN: one register
    const float pix_x0 = (float)gx, pix_y0 = (float)gy;
    for (;;) {
        if (COUNT && first_active_lane()) wc->wtrips++;
        if (fresh && !traversing) {            // a lane whose path ended starts its next sample right here
            if (s == s_end) break;
            st.reset();                        // prog.cl:307-316
            inside = false;
            if (SPLIT) {
                const float4* r = reinterpret_cast<const float4*>(&p.rays[px.li]);
                const float4 a = r[0], b = r[1];
                rP = mk(a.x, a.y, a.z);
                rD = mk(b.x, b.y, b.z);
            } else {
                const float rnd1 = lcg_rand(seed), rnd2 = lcg_rand(seed);
                const float pix_x = LEAN ? (float)(pxy & 0xffffu) : pix_x0, pix_y = LEAN ? (float)(pxy >> 16) : pix_y0;
                camera_get_ray_xy(pix_x, pix_y, p.cam, rnd1, rnd2, &rP, &rD);
            }
            bounce = 0;
            fresh = false;
        }
        bool finished = true;
        if (traversing || bounce < p.iterations) {
            tr.setup(rP, rD);                  // direction-dependent constants: recomputed for new and resumed rays alike
            if (!traversing) {
                tr.restart(stk);
                tr.template flat_pass<COUNT>(sv, wc);
            }
            for (;;) {
                if (COUNT && first_active_lane()) wc->wrounds++;
                tr.template round<COUNT>(sv, wc);
                const unsigned long long unfinished = __ballot(!tr.done());
                if (unfinished == 0) break;
                if (__popcll(unfinished) <= p.suspend_lanes && __ballot(tr.done()) != 0) break;
            }
            traversing = !tr.done();
            if (traversing) {
                finished = false;
            } else {
                ++*segs;
                if (tr.best >= 0) {
                    if (COUNT) { if (first_active_lane()) wc->wshade++; count_low(wc, 3); }
                    shade_hit<SK>(rP, rD, st, seed, inside, p, sv.tris, sv.meta, tr.best, tr.best_t);
                    ++bounce;
                    finished = (bounce >= p.iterations);
                }
            }
        }
        if (finished) {
            if (LEAN) fold_sample(p, px.li, st.C(), s);
            else acc = running_mean(acc, st.C(), s);
            ++s;
                fresh = true;
        }
    }
    if (!LEAN) p.colors[px.li] = make_float4(acc.x, acc.y, acc.z, 0.0f);
    p.rnds[px.li] = seed;
    if (SPLIT) {
        float4* r = reinterpret_cast<float4*>(&p.rays[px.li]);
        r[0] = make_float4(rP.x, rP.y, rP.z, 0.0f);
        r[1] = make_float4(rD.x, rD.y, rD.z, 0.0f);
    }
}

// Chained passes: wait until pass `pass` - 1 of `tile` has been released (tile_done[tile] >= pass).  The producer was dequeued earlier
// by a resident wave, so the wait ends -- unless that wave died (a fault in its item) or the launch is already winding down.  The poll
// is therefore BOUNDED: after poll_ticks (10 s by default) the wave reports the tile in tile_counter[kTileCounterError] and gives up;
// every other waiting wave sees that word and gives up too, the launch drains, and the host turns the word into PT_EHIP (pt_sync)
// instead of a hung process.  Returns 1 when the hand-over came, 0 when it was lost (here or elsewhere).  Wave-uniform throughout.
PT_DEV int wait_tile_pass(const RenderParams& p, int tile, int pass, bool lane0) {
    unsigned seen = 0, lost = 0, polls = 0;
    const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        if (lane0) seen = __hip_atomic_load(&p.tile_done[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);
        if (seen >= (unsigned)pass) break;
        // now and then (every 64th poll: thousands of waves poll at once, and one word read by all of them every
        // microsecond is a hot spot of its own): has another wave given up, or is it time to?
        if ((++polls & 63u) == 0) {
            if (lane0) lost = __hip_atomic_load(&p.tile_counter[kTileCounterError], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lost = (unsigned)__builtin_amdgcn_readfirstlane((int)lost);
            if (lost != 0) break;
            if (__builtin_amdgcn_s_memrealtime() - w0 > (unsigned long long)p.poll_ticks) {
                // (tile and pass in ONE 8-byte compare-and-swap: the first wave to give up names them)
                const unsigned long long mine = ((unsigned long long)(unsigned)pass << 32) | (unsigned long long)((unsigned)tile + 1u);
                unsigned long long was = 0;
                if (lane0) was = atomicCAS(reinterpret_cast<unsigned long long*>(&p.tile_counter[kTileCounterError]), 0ull, mine);
                (void)__builtin_amdgcn_readfirstlane((int)(unsigned)was);
                break;
            }
        }
        __builtin_amdgcn_s_sleep(8);
    }
    return __builtin_amdgcn_readfirstlane((int)(seen >= (unsigned)pass));
}

// kSchedMigrate: the whole item loop of a persistent launch (p.tile_counter != 0), chained passes or whole tiles.
//
// Wave-uniform: the current item (cur_*) and, once a lane has finished its pixel of it, the next one (nxt_*), fetched from the work
// counter like every item.  Per lane: the pixel in hand (li, pxy, seed, running mean), whether it belongs to the next item (on_next),
// and `parked` = no pixel in hand (finished the item's pixel, or the tile has no pixel for this lane).
//   * a lane whose pixel is done parks; parked lanes of the CURRENT item are moved to the next item as soon as that item may start;
//   * a chained item may start when its tile's previous pass has been released.  While lanes still work on the current item that is
//     only LOOKED AT (one relaxed load per trip) -- the wave never spins with unfinished work in hand, so the argument of k_render
//     holds as it stands: a wave blocks only with every earlier item of its own released, on an item that was dequeued before;
//   * when no lane works on the current item any more it is released, and the next item becomes the current one (lanes that have
//     finished THAT one already stay parked until the item after it is fetched).
// Results are those of the other schedules bit for bit: what a pixel computes never depends on which lanes run next to it.
template <int MODE, bool COUNT, bool LEAN, bool SK>
PT_DEV void render_items_migrating(const RenderParams& p, const SceneView& sv, const LaneStack<typename StackOf<MODE>::type> stk, WorkCount* wc,
                                   unsigned long long* segs_tot, unsigned long long* samples_tot, unsigned long long* lane_steps) {
    const bool lane0 = (threadIdx.x & 63) == 0;
    const bool chained = p.chunk_spp > 0;
    const int n_pass = chained ? (p.n_taper > 0 ? p.n_taper : (p.nsamples + p.chunk_spp - 1) / p.chunk_spp) : 1;
    const int s_last = p.first_sample + p.nsamples;
    const int camX = (int)p.cam.XM;
    // ---- per lane
    f3 rP = mk(0.f, 0.f, 0.f), rD = mk(0.f, 0.f, 1.f);
    PathRegs st;
    st.reset();
    bool inside = false;
    int seed = 0;
    f3 acc = mk(0.0f, 0.0f, 0.0f);
    int s = 0, bounce = 0;
    bool fresh = true, traversing = false;
    bool parked = true, on_next = false;
    int li = 0;
    unsigned pxy = 0;
    unsigned my_segs = 0;
    Trav<MODE> tr;
    tr.setup(rP, rD);
    tr.restart(stk);
    tr.idle();
    // ---- per wave (scalar registers: readfirstlane at every redefinition, as in k_render).  An item is the number the work
    // counter gave (pass * n_tiles + tile; -1: none): two registers for the two items in hand, everything else is derived
    int cur_item = -1, nxt_item = -1;
    int cur_end = 0, nxt_end = 0;          // where their samples end
    int nxt_ready = 0, no_more = 0;
    unsigned long long samples = 0;
    auto pass_of = [&](int item) { return chained ? item / p.n_tiles : 0; };
    auto tile_of = [&](int item) { return chained ? item - (item / p.n_tiles) * p.n_tiles : item; };

    // the next work item off the counter; -1: none left (every wave gets exactly one such answer: k_render's exit protocol)
    auto fetch = [&]() {
        int t = 0;
        if (lane0) t = (int)atomicAdd(p.tile_counter, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        const bool ok = chained ? t / p.n_tiles < n_pass : t < p.n_tiles;
        return __builtin_amdgcn_readfirstlane(ok ? t : -1);
    };
    auto released = [&](int item) {        // has the tile's previous pass been handed over?  (one look)
        const int pass = pass_of(item);
        if (pass == 0) return 1;
        unsigned seen = 0;
        if (lane0) seen = __hip_atomic_load(&p.tile_done[tile_of(item)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);
        return __builtin_amdgcn_readfirstlane((int)(seen >= (unsigned)pass));
    };
    auto wait_released = [&](int item) {       // ... or wait for it (bounded: wait_tile_pass)
        const int pass = pass_of(item);
        return pass == 0 ? 1 : wait_tile_pass(p, tile_of(item), pass, lane0);
    };
    auto begin_of = [&](int item) { const int ps = pass_of(item); return p.first_sample + (p.n_taper > 0 ? (ps > 0 ? (int)p.taper_end[ps - 1] : 0) : ps * p.chunk_spp); };
    auto end_of = [&](int item) { return !chained ? s_last : p.n_taper > 0 ? p.first_sample + (int)p.taper_end[pass_of(item)] : min(p.first_sample + (pass_of(item) + 1) * p.chunk_spp, s_last); };
    // this lane's pixel of an item: LCG state, running mean, first sample
    // this lane's pixel of an item: LCG state, running mean, first sample.  A lane comes here on its own, a few at a time, so
    // what pixel_of_wave() and prog.cl:84-85 spend on integer divisions per lane is done on the scalar unit where the frame
    // allows it (tile rows never straddle a block of rows_per_block rows when that is a multiple of 8; camera as wide as the frame)
    const int lane = threadIdx.x & 63;
    const int tiles_x = (p.width + 7) >> 3;
    const bool rows_aligned = (p.rows_per_block & 7) == 0;
    auto take_pixel = [&](int item) {
        const int tile = tile_of(item);
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;            // (wave-uniform: scalar unit)
        const int x = tx * 8 + (lane & 7), lrow = ty * 8 + (lane >> 3);
        if (x >= p.width || lrow >= p.local_rows) return;                   // (stays parked: the tile has no pixel for this lane)
        int grow;
        if (p.world == 1) grow = lrow;
        else if (rows_aligned) { const int r0 = ty * 8; grow = ((r0 / p.rows_per_block) * p.world + p.rank) * p.rows_per_block + (r0 % p.rows_per_block) + (lane >> 3); }
        else grow = ((lrow / p.rows_per_block) * p.world + p.rank) * p.rows_per_block + (lrow % p.rows_per_block);
        li = lrow * p.width + x;
        int gx = x, gy = grow;                                               // prog.cl:84-85: id % X, id / X with id = grow * width + x
        if (camX != p.width) { const int gid = grow * p.width + x; gx = gid % camX; gy = gid / camX; }
        pxy = (unsigned)gx | ((unsigned)gy << 16);
        seed = p.rnds[li];
        s = begin_of(item);
        if (!LEAN) {
            acc = mk(0.0f, 0.0f, 0.0f);
            if (s != 0) {                          // prog.cl:312-314: sample 0 starts from black
                const float4 c = p.colors[li];
                acc = mk(c.x, c.y, c.z);
            }
        }
        fresh = true;
        traversing = false;
        bounce = 0;
        parked = false;
    };

    for (;;) {
        if (COUNT && first_active_lane()) wc->wtrips++;
        // ---- a pixel that has had its samples of the item is written out
        if (!parked && fresh && !traversing && s == (on_next ? nxt_end : cur_end)) {
            if (!LEAN) p.colors[li] = make_float4(acc.x, acc.y, acc.z, 0.0f);
            p.rnds[li] = seed;
            parked = true;
        }
        // ---- the wave's items
        if (__ballot(!parked && !on_next) == 0) {
            // nobody works on the current item any more: hand it over, the next one becomes the current one
            if (cur_item >= 0 && chained) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // keep the wait the compiler may drop (guide, G16)
                const int tile = tile_of(cur_item), pass = pass_of(cur_item);
                if (lane0 && !(tile == p.debug_stall_tile && pass == 0)) {
                    unsigned handed;             // (the number comes out of a scalar register right here: tools/check_isa.py, rule 2)
                    asm volatile("v_mov_b32 %0, %1" : "=v"(handed) : "s"(__builtin_amdgcn_readfirstlane(pass + 1)));
                    __hip_atomic_store(&p.tile_done[tile], handed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            cur_item = -1;
            if (nxt_item < 0 && !no_more) {
                nxt_item = fetch();
                no_more = __builtin_amdgcn_readfirstlane((int)(nxt_item < 0));
                nxt_ready = 0;
            }
            if (nxt_item < 0) break;                                 // no item left, nothing in hand
            if (!nxt_ready) {
                if (!wait_released(nxt_item)) break;                 // a hand-over was lost somewhere: this wave renders nothing more
                if (pass_of(nxt_item) > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            cur_item = __builtin_amdgcn_readfirstlane(nxt_item);
            cur_end = __builtin_amdgcn_readfirstlane(end_of(nxt_item));
            nxt_item = -1;
            nxt_ready = 0;
            if (on_next) on_next = false;                            // already at work on it (or done with it: stays parked)
            else take_pixel(cur_item);                               // (every lane that was not on it is parked here)
        } else if (__popcll(__ballot(parked && !on_next)) >= p.migrate_lanes) {
            // lanes are done with the current item while others still work on it: start them on the next one, if it may start
            // (migrate_lanes = how many must have gathered: moving one by one measured best, profiles/r04/r_*)
            if (nxt_item < 0 && !no_more) {
                nxt_item = fetch();
                no_more = __builtin_amdgcn_readfirstlane((int)(nxt_item < 0));
                nxt_ready = 0;
                nxt_end = __builtin_amdgcn_readfirstlane(end_of(nxt_item));
            }
            if (nxt_item >= 0 && !nxt_ready && released(nxt_item)) {
                if (pass_of(nxt_item) > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                nxt_ready = 1;
            }
            if (nxt_item >= 0 && nxt_ready && parked && !on_next) {
                on_next = true;
                take_pixel(nxt_item);
            }
        }
        // ---- the segment loop of kSchedSuspend for the lanes with a pixel in hand
        if (!parked && fresh && !traversing) {       // a lane whose path ended starts its next sample right here
            st.reset();                              // prog.cl:307-316
            inside = false;
            const float rnd1 = lcg_rand(seed), rnd2 = lcg_rand(seed);
            camera_get_ray_xy((float)(pxy & 0xffffu), (float)(pxy >> 16), p.cam, rnd1, rnd2, &rP, &rD);
            bounce = 0;
            fresh = false;
        }
        bool finished = !parked;
        if (!parked && (traversing || bounce < p.iterations)) {
            tr.setup(rP, rD);                        // direction-dependent constants: recomputed for new and resumed rays alike
            if (!traversing) {
                tr.restart(stk);
                tr.template flat_pass<COUNT>(sv, wc);
            }
            for (;;) {
                if (COUNT && first_active_lane()) wc->wrounds++;
                tr.template round<COUNT>(sv, wc);
                const unsigned long long unfinished = __ballot(!tr.done());
                if (unfinished == 0) break;
                if (__popcll(unfinished) <= p.suspend_lanes && __ballot(tr.done()) != 0) break;
            }
            traversing = !tr.done();
            if (traversing) {
                finished = false;
            } else {
                ++my_segs;
                if (tr.best >= 0) {
                    if (COUNT) { if (first_active_lane()) wc->wshade++; count_low(wc, 3); }
                    shade_hit<SK>(rP, rD, st, seed, inside, p, sv.tris, sv.meta, tr.best, tr.best_t);
                    ++bounce;
                    finished = (bounce >= p.iterations);
                }
            }
        }
        samples += (unsigned long long)__popcll(__ballot(finished));
        if (finished) {
            if (LEAN) fold_sample(p, li, st.C(), s);
            else acc = running_mean(acc, st.C(), s);
            ++s;
            fresh = true;
        }
    }
    {
        unsigned v = my_segs;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        *segs_tot += (unsigned)__builtin_amdgcn_readfirstlane((int)v);
        *samples_tot += samples;
    }
    if (COUNT) {      // segment-steps of the wave = 64 x its busiest lane's segments (over all its items: lanes move on their own)
        unsigned mx = my_segs;
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_down(mx, off, 64));
        *lane_steps += (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)mx) * 64ull;
    }
}

// The render kernel.  One wave = one 8x8 pixel tile; each lane owns one pixel.
//
// Launch modes:
//  * plain (p.tile_counter == 0): wave w of the grid renders tile w, all samples.
//  * persistent (p.tile_counter != 0): the grid only fills the machine and every WAVE pulls its next
//    work item from a global counter as soon as it is done -- it never waits for the other waves of
//    its workgroup, whose tiles may take 30 % longer (tiles over the spheres vs. bare walls).
//  * persistent with chained passes (p.chunk_spp > 0): a work item is (pass, tile) = chunk_spp
//    samples of one tile, handed out pass-major.  The producer of a tile's previous pass was dequeued
//    n_tiles items earlier by a resident wave, so waiting for it cannot deadlock; its rnds/colors
//    reach this wave -- possibly on another XCD -- through an agent-scope release (producer: stores,
//    L2 write-back, tile_done[tile] = pass + 1) and acquire (consumer: poll, L1 invalidate, loads).
//    The host guarantees n_pass * n_tiles + (resident waves) < 2^31 (pt_render).
// WPS = waves per SIMD the register budget is set for (512 / WPS VGPRs): 4 where the kernel is bound by VALU
// issue (nodes in LDS); 5, 6 or 7 for nodes from global memory, where every wave-level step is a dependent memory
// round trip and another resident wave hides more of it than the extra spills cost (4 -> 5 -> 6 waves: MESH-100k
// 630 / 698 / 727 with BVH2 nodes, a seventh +1-2 %, an eighth loses; the step from 4 to 5 lost 30 % on the Cornell
// box: profiles/r02/t_*, v_*).
template <bool SPLIT, int MODE, int BLOCK, bool COUNT, int SCHED, int WPS>
__global__ void __launch_bounds__(BLOCK, WPS) k_render(RenderParams p) {
    constexpr bool kLean = WPS >= kLeanFromWps;      // see render_pixel_*
    constexpr bool kScalarK = WPS >= 5;              // double-precision constants in scalar registers (KC, pt_device.hpp)
    LaneStack<typename StackOf<MODE>::type> stk;
    SceneView sv;
    setup_traversal<MODE, BLOCK>(p, &sv, &stk);
    WorkCount wc;
    wc.nodes = 0;
    wc.tris = 0;
    wc.wnodes = 0;
    wc.wtris = 0;
    wc.wshade = 0;
    wc.wtrips = 0;
    wc.wrounds = 0;
    for (int i = 0; i < 6; ++i) wc.low[i] = 0;
    const bool lane0 = (threadIdx.x & 63) == 0;
    const bool chained = p.tile_counter != nullptr && p.chunk_spp > 0;
    const int n_pass = chained ? (p.n_taper > 0 ? p.n_taper : (p.nsamples + p.chunk_spp - 1) / p.chunk_spp) : 1;
    // statistics of the launch: wave-level totals, kept in SCALAR registers (a per-lane 64-bit counter pair costs four VGPRs
    // through every traversal); a lane only counts the segments of its current work item, in 32 bits
    unsigned long long segs_tot = 0, samples_tot = 0, item_lane_steps = 0;
    int tile = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    int pass = 0;
    if (SCHED == kSchedMigrate) {                                // (persistent launches only: launch_render_mega)
        if (p.tile_counter) render_items_migrating<MODE, COUNT, kLean, kScalarK>(p, sv, stk, &wc, &segs_tot, &samples_tot, &item_lane_steps);
    } else
    for (;;) {
        if (p.tile_counter) {                                    // ---- fetch the next work item
            // (the work item is wave-uniform: kept in scalar registers -- readfirstlane, not a lane shuffle -- so that
            // it is never spilled lane by lane under whatever exec mask the spill happens to land in)
            int t = 0;
            if (lane0) t = (int)atomicAdd(p.tile_counter, 1u);
#ifdef PT_TEST_WORK_ITEM_IN_VGPR      // round 2's bug, kept only to show that tools/check_isa.py catches it (never built into the library)
            t = __shfl(t, 0, 64);
#else
            t = __builtin_amdgcn_readfirstlane(t);
#endif
            pass = chained ? t / p.n_tiles : 0;
            tile = t - pass * p.n_tiles;
            if (chained ? pass >= n_pass : tile >= p.n_tiles) break;
            if (pass > 0) {                                      // ---- acquire the tile's previous pass
                if (!wait_tile_pass(p, tile, pass, lane0)) break;   // a hand-over was lost somewhere: this wave renders nothing more
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        const bool tabled = chained && p.n_taper > 0;           // (passes of different lengths: RenderParams::taper_end)
        const int s_begin = p.first_sample + (tabled ? (pass > 0 ? (int)p.taper_end[pass - 1] : 0) : chained ? pass * p.chunk_spp : 0);
        const int s_end = tabled ? p.first_sample + (int)p.taper_end[pass] : chained ? min(s_begin + p.chunk_spp, p.first_sample + p.nsamples) : p.first_sample + p.nsamples;
        const PixelId px = pixel_of_wave(p, tile);
        unsigned item_segs = 0;
        const unsigned long long item_t0 = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
        if (px.li >= 0) {
            if (SCHED == kSchedLockstep) render_pixel_lockstep<SPLIT, MODE, COUNT, kLean, kScalarK>(p, sv, stk, px, s_begin, s_end, &item_segs, &wc);
            else render_pixel_suspend<SPLIT, MODE, COUNT, kLean, kScalarK>(p, sv, stk, px, s_begin, s_end, &item_segs, &wc);
        }
        {
            unsigned v = item_segs;
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            segs_tot += (unsigned)__builtin_amdgcn_readfirstlane((int)v);
            samples_tot += (unsigned long long)__popcll(__ballot(px.li >= 0)) * (unsigned)(s_end - s_begin);
        }
        if (COUNT) {      // segment-steps the wave executed for this item = 64 x the busiest lane's segments
            unsigned mx = item_segs;
            for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_down(mx, off, 64));
            const unsigned mxs = (unsigned)__builtin_amdgcn_readfirstlane((int)mx);
            item_lane_steps += (unsigned long long)mxs * 64ull;
            // what the tile cost THIS wave, in shader-clock cycles / 64 (the wave shares its SIMD with the others resident there:
            // that is the latency a launch with one tile per wave ends on)
            if (p.tile_cost && lane0) atomicAdd(&p.tile_cost[tile], (unsigned)((__builtin_amdgcn_s_memtime() - item_t0) >> 6));
        }
        if (chained) {                                           // ---- release this pass of the tile
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // keep the wait the compiler may drop (guide, G16)
            if (lane0 && !(tile == p.debug_stall_tile && pass == 0))
                __hip_atomic_store(&p.tile_done[tile], (unsigned)(pass + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!p.tile_counter) break;
    }
    if (p.tile_counter) {
        // Last wave out resets the work counter (word 0) and the exit count (word 1), so the host never clears them: every
        // wave of the grid comes here exactly once, after its one failed fetch, and nobody reads word 0 after that.
        unsigned gone = 0;
        if (lane0) gone = atomicAdd(p.tile_counter + 1, 1u);
        gone = (unsigned)__builtin_amdgcn_readfirstlane((int)gone);
        if (lane0 && gone + 1u == gridDim.x * (BLOCK / 64)) {
            __hip_atomic_store(p.tile_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.tile_counter + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (COUNT) {
        const unsigned long long wn = wave_sum((unsigned long long)wc.nodes), wt = wave_sum((unsigned long long)wc.tris);
        const unsigned long long wwn = wave_sum((unsigned long long)wc.wnodes), wwt = wave_sum((unsigned long long)wc.wtris);
        const unsigned long long wsh = wave_sum((unsigned long long)wc.wshade), wtr = wave_sum((unsigned long long)wc.wtrips), wro = wave_sum((unsigned long long)wc.wrounds);
        if (lane0 && p.stats) {
            stat_add(p, 2, wn);
            stat_add(p, 3, wt);
            stat_add(p, 4, wwn);
            stat_add(p, 5, wwt);
            stat_add(p, 6, item_lane_steps);
            stat_add(p, 7, wsh);
            stat_add(p, 8, wtr);
            stat_add(p, 9, wro);
        }
        for (int i = 0; i < 6; ++i) {
            const unsigned long long v = wave_sum((unsigned long long)wc.low[i]);
            if (lane0 && p.stats) stat_add(p, 10 + i, v);
        }
    }
    if (lane0 && p.stats) {
        stat_add(p, 0, segs_tot);
        stat_add(p, 1, samples_tot);
    }
}

// ---- tone mapping, prog.cl:247-269 (value of write_imagef at prog.cl:380)
PT_DEV float srgb1(float a) {
    if (a <= 0.00304f) return 12.92f * a;
    return fmaf_(1.055f, spec_pow<false>(a, 0.4167f), -0.055f);
}
__global__ void __launch_bounds__(256) k_resolve_reinhard(const float4* colors, float4* out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = colors[i];
    const float L = fmaf_(0.0722f, c.z, fmaf_(0.7152f, c.y, 0.2126f * c.x));
    const float L2 = L / (1.0f + L);
    out[i] = make_float4(srgb1(c.x * L2 / L), srgb1(c.y * L2 / L), srgb1(c.z * L2 / L), 1.0f);
}

// filt_im, prog.cl:391-427: 3x3 median by mean grey + filmic tone map; the reference's
// out-of-range reads at the right/top border (prog.cl:397-401) are not reproduced: border
// pixels are left untouched.
PT_DEV float filmic1(float cin) {   // prog.cl:259-263
    float c = cin - 0.004f;
    c = c > 0.0f ? c : 0.0f;
    return (c * fmaf_(c, 6.2f, 0.5f)) / fmaf_(c, fmaf_(c, 6.2f, 1.7f), 0.06f);
}
__global__ void __launch_bounds__(256) k_filt_im(const float4* colors, float4* out, int W, int H) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x <= 0 || y <= 0 || x >= W - 1 || y >= H - 1) return;
    float4 arr[9];
    float grey[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float4 c = colors[(size_t)(y - 1 + i) * W + (x - 1 + j)];
            arr[i * 3 + j] = c;
            grey[i * 3 + j] = ((c.x + c.y) + c.z) / 3.0f;
        }
#pragma unroll
    for (int jn = 9; jn > 1; --jn) {
        int maxi = 0;
#pragma unroll
        for (int i = 1; i < jn; ++i)
            if (grey[i] > grey[maxi]) maxi = i;
#pragma unroll
        for (int i = 0; i < 9; ++i) {       // swap without dynamic register indexing
            if (i == maxi) {
                const float tg = grey[jn - 1];
                const float4 tv = arr[jn - 1];
                grey[jn - 1] = grey[i];
                arr[jn - 1] = arr[i];
                grey[i] = tg;
                arr[i] = tv;
            }
        }
    }
    out[(size_t)y * W + x] = make_float4(filmic1(arr[4].x), filmic1(arr[4].y), filmic1(arr[4].z), 1.0f);
}

// ---------------------------------------------------------------------------- launchers
static inline int n_waves(const RenderParams& p) {
    const int tiles_x = (p.width + 7) >> 3, tiles_y = (p.local_rows + 7) >> 3;
    return tiles_x * tiles_y;
}

int traversal_block(int node_mode, bool wide_lds_block) { return node_mode == kNodesLds ? (wide_lds_block ? kLdsBlockWide : kLdsBlockBase) : node_mode == kNodesTreelet ? 1024 : 256; }

size_t traversal_lds_bytes(const RenderParams& p, int block) {
    size_t b = (size_t)p.stack_entries * (p.node_mode == kNodesLds ? 2 : 4) * (size_t)block;
    b = (b + 15) & ~(size_t)15;
    if (p.node_mode == kNodesLds) b += ((size_t)p.n_nodes * kLdsNodeBytes + 15) & ~(size_t)15;
    if (p.node_mode == kNodesTreelet) b += (size_t)p.treelet_nodes * 64;
    return b + (size_t)p.n_flat * 100;         // the big-triangle list's packets, padded boxes and box-group masks
}

hipError_t launch_gen_ray(const RenderParams& p, hipStream_t stream) {
    const int waves = n_waves(p);
    if (waves == 0) return hipSuccess;
    const int blocks = (waves + 3) / 4;
    hipLaunchKernelGGL(k_gen_ray, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <bool SPLIT, int MODE, int BLOCK, bool COUNT, int SCHED, int WPS>
static hipError_t launch_one(const RenderParams& p, const LaunchConfig& lc, hipStream_t stream) {
    const int waves = n_waves(p);
    if (waves == 0) return hipSuccess;
    constexpr int wpb = BLOCK / 64;
    int blocks = (waves + wpb - 1) / wpb;
    if (p.tile_counter) blocks = std::min(blocks, lc.persistent_blocks);
    if (p.stack_ovf && (long long)blocks * BLOCK > (long long)p.stack_ovf_lanes) return hipErrorInvalidValue;   // (pt_launch.cpp alloc_stack_overflow)
    auto kern = k_render<SPLIT, MODE, BLOCK, COUNT, SCHED, WPS>;
    static LdsMark mark;
    const hipError_t e = ensure_dynamic_lds((const void*)kern, mark, lc.lds_bytes);
    if (e != hipSuccess) return e;
    if (p.tile_counter && lc.cu_count > 0) {         // a persistent grid: never more workgroups than the runtime says are co-resident
        static OccMark occ;
        const int per_cu = resident_blocks_per_cu((const void*)kern, occ, BLOCK, lc.lds_bytes);
        if (per_cu > 0) blocks = std::min(blocks, per_cu * lc.cu_count);
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(BLOCK), lc.lds_bytes, stream, p);
    return hipGetLastError();
}

template <bool SPLIT, bool COUNT, int SCHED>
static hipError_t launch_render_t(const RenderParams& p, const LaunchConfig& lc, hipStream_t stream) {
    if (lc.block != traversal_block(p.node_mode, lc.block == kLdsBlockWide)) return hipErrorInvalidValue;
    switch (p.node_mode) {
    case kNodesLds:
        if (lc.block == kLdsBlockWide) return launch_one<SPLIT, kNodesLds, kLdsBlockWide, COUNT, SCHED, kLdsWpsWide>(p, lc, stream);
        return launch_one<SPLIT, kNodesLds, kLdsBlockBase, COUNT, SCHED, kLdsWpsBase>(p, lc, stream);
    case kNodesGlobal:
        if (lc.waves_per_simd == 8) return launch_one<SPLIT, kNodesGlobal, 256, COUNT, SCHED, 8>(p, lc, stream);
        if (lc.waves_per_simd == 7) return launch_one<SPLIT, kNodesGlobal, 256, COUNT, SCHED, 7>(p, lc, stream);
        if (lc.waves_per_simd == 6) return launch_one<SPLIT, kNodesGlobal, 256, COUNT, SCHED, 6>(p, lc, stream);
        if (lc.waves_per_simd == 5) return launch_one<SPLIT, kNodesGlobal, 256, COUNT, SCHED, 5>(p, lc, stream);
        return launch_one<SPLIT, kNodesGlobal, 256, COUNT, SCHED, 4>(p, lc, stream);
    case kNodesWide:
        if (lc.waves_per_simd == 8) return launch_one<SPLIT, kNodesWide, 256, COUNT, SCHED, 8>(p, lc, stream);
        if (lc.waves_per_simd == 7) return launch_one<SPLIT, kNodesWide, 256, COUNT, SCHED, 7>(p, lc, stream);
        if (lc.waves_per_simd == 6) return launch_one<SPLIT, kNodesWide, 256, COUNT, SCHED, 6>(p, lc, stream);
        if (lc.waves_per_simd == 5) return launch_one<SPLIT, kNodesWide, 256, COUNT, SCHED, 5>(p, lc, stream);
        return launch_one<SPLIT, kNodesWide, 256, COUNT, SCHED, 4>(p, lc, stream);
    case kNodesTreelet: return launch_one<SPLIT, kNodesTreelet, 1024, COUNT, SCHED, 4>(p, lc, stream);
    }
    return hipErrorInvalidValue;
}

// The split API traces one sample per launch: the two schedules coincide, one instance suffices.  (Round 3 tried a third
// one for it -- a wave streaming through its tiles lane by lane, each lane moving on the moment its own path has ended:
// 1,415 against 1,784 Msamples/s.  With one sample per pixel the camera rays of a tile traced TOGETHER are worth more than
// the idle lanes cost: profiles/r03/i_*.)
hipError_t launch_trace_ray(const RenderParams& p, const LaunchConfig& lc, hipStream_t stream) { return launch_render_t<true, false, kSchedLockstep>(p, lc, stream); }
hipError_t launch_render_mega(const RenderParams& p, const LaunchConfig& lc, hipStream_t stream) {
    if (lc.schedule == kSchedLockstep)
        return lc.count_work ? launch_render_t<false, true, kSchedLockstep>(p, lc, stream) : launch_render_t<false, false, kSchedLockstep>(p, lc, stream);
    if (lc.schedule == kSchedMigrate && p.tile_counter)
        return lc.count_work ? launch_render_t<false, true, kSchedMigrate>(p, lc, stream) : launch_render_t<false, false, kSchedMigrate>(p, lc, stream);
    return lc.count_work ? launch_render_t<false, true, kSchedSuspend>(p, lc, stream) : launch_render_t<false, false, kSchedSuspend>(p, lc, stream);
}

hipError_t launch_resolve_reinhard(const float4* colors, float4* out, int64_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_resolve_reinhard, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, colors, out, (long long)n);
    return hipGetLastError();
}

hipError_t launch_filt_im(const float4* colors, float4* out, int32_t W, int32_t H, hipStream_t stream) {
    hipLaunchKernelGGL(k_filt_im, dim3((W + 31) / 32, (H + 7) / 8), dim3(256), 0, stream, colors, out, W, H);
    return hipGetLastError();
}

}  // namespace ptamd
