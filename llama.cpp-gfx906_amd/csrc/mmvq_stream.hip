// mmvq_stream.hip — host side of the streamed mat-vec (device code and the design: mmvq_stream.h): decides whether a grouped launch can
// take this path, deals the workgroups (one per CU) to the groups in proportion to their weight bytes, sizes the LDS carve and the slot
// ring, and launches. Called by mul_mat_vec_q_fused (mmvq_fused.hip), which falls back to round 2's register-ring kernels when this
// returns false (MUL_MAT_ID expert stacks, Q4_0 / Q8_0 / MXFP4 weights, rows longer than 16384 elements, strided rows).
#include <hip/hip_ext.h>

#include "mmvq_stream.h"

#include <limits.h>
#include <string.h>
#include <algorithm>

namespace mi355x {

template <int TA, int TB, bool GLU> static void st_launch_t(const st_args & a, int blocks, size_t lds, hipStream_t stream, hipEvent_t e0, hipEvent_t e1, const char ** kname) {
    static char name[64] = "";
    if (!name[0]) snprintf(name, sizeof(name), "k_mmvq_stream<%d, %d, true, %s>", TA, TB, GLU ? "true" : "false");
    *kname = name;
    MI_LDS_LIMIT_OR_DIE(163840, k_mmvq_stream<TA, TB, true, GLU>);
    if (e0) hipExtLaunchKernelGGL((k_mmvq_stream<TA, TB, true, GLU>), dim3((unsigned) blocks), dim3(ST_THREADS), lds, stream, e0, e1, 0, a);
    else    hipLaunchKernelGGL((k_mmvq_stream<TA, TB, true, GLU>), dim3((unsigned) blocks), dim3(ST_THREADS), lds, stream, a);
}

static int st_unit_bytes(int type) { return type == T_Q4_K ? 144 : type == T_Q5_K ? 176 : type == T_Q6_K ? 210 : type == T_Q8_0 ? 272 : type == T_Q4_0 ? 144 : type == ST_Q8_0_B10 ? 340 : type == ST_MXFP4_B10 ? 170 : 0; }
// the unit a weight type is streamed in at row length k: 256 weights, or ten 32-blocks where k is not a multiple of 256 (gpt-oss: 2880); 0: none
static int st_utype(int type, int64_t k) {
    if (k % 256 == 0) return type == T_MXFP4 ? 0 : type;
    if (k % 320 == 0 && type == T_Q8_0) return ST_Q8_0_B10;
    if (k % 320 == 0 && type == T_MXFP4) return ST_MXFP4_B10;
    return 0;
}
static int st_unit_weights(int utype) { return utype == ST_Q8_0_B10 || utype == ST_MXFP4_B10 ? 320 : 256; }
static int st_cu_count() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    return n_cu;
}

#ifdef MI_STAMPS
static unsigned long long * g_st_stamps = nullptr; static int g_st_slots = 0, g_st_next = 0;
struct st_stamp_meta { int blocks, k, rows, type_a, type_b, mode, glu; long long bytes; };
static st_stamp_meta g_st_meta[4096];
extern "C" int mi355x_stream_stamps_enable(int n_slots) {
    if (g_st_stamps) { (void) hipFree(g_st_stamps); g_st_stamps = nullptr; }
    g_st_slots = n_slots > 4096 ? 4096 : n_slots; g_st_next = 0;
    if (g_st_slots <= 0) return 0;
    const size_t bytes = (size_t) g_st_slots*256*(ST_NC + 1)*ST_NSTAMP*8;
    if (hipMalloc(&g_st_stamps, bytes) != hipSuccess) return -1;
    (void) hipMemset(g_st_stamps, 0, bytes);
    return 0;
}
extern "C" int mi355x_stream_stamps_used(void) { return g_st_next; }
extern "C" int mi355x_stream_stamps_read(int slot, unsigned long long * out, int * meta, long long * bytes) {
    if (!g_st_stamps || slot < 0 || slot >= g_st_slots) return -1;
    const st_stamp_meta & m = g_st_meta[slot];
    (void) hipMemcpy(out, g_st_stamps + (size_t) slot*256*(ST_NC + 1)*ST_NSTAMP, (size_t) 256*(ST_NC + 1)*ST_NSTAMP*8, hipMemcpyDeviceToHost);
    meta[0] = m.blocks; meta[1] = m.k; meta[2] = m.rows; meta[3] = m.type_a; meta[4] = m.type_b; meta[5] = m.mode; meta[6] = m.glu;
    *bytes = m.bytes;
    return 0;
}
#endif

bool mul_mat_vec_q_stream_enabled(void) {
    static int on = -1;
    if (on < 0) { const char * e = getenv("GGML_MI355X_STREAM"); on = e ? atoi(e) : 1; }
    return on != 0;
}

static int st_fill(const mmvq_group * groups, int n_groups, int64_t k, const mmvq_input & in, const mmvq_rope * rope, st_args & a,
                   size_t & fixed_max, int & slot_max, int & nslots_max, int & npart_max, int & ta, int & tb, double & bytes_total);

// can (and does) this grouped launch go to the streamed kernel?
bool mul_mat_vec_q_stream_takes(const mmvq_group * groups, int n_groups, int64_t k, const mmvq_input & in, const mmvq_rope * rope) {
    if (!mul_mat_vec_q_stream_enabled() || n_groups < 1 || n_groups > MMVQ_MAX_GROUPS) return false;
    if ((k % 256 != 0 && k % 320 != 0) || k > 16384 || k < 256) return false;
    const bool b10 = k % 256 != 0;         // ten-block units (Q8_0 / MXFP4 rows of 320 n weights)
    if (b10 && in.mode == PRO_Q8) return false;
    if (in.mode != PRO_Q8 && in.mode != PRO_QUANT && in.mode != PRO_NORM) return false;
    if (in.mode == PRO_Q8 && in.act_kind != T_Q8_K) return false;      // (a ready-made image: Q8_K only; the launch's own prologue quantizes per workgroup, in the format its group's weights ask for)
    if (in.mode == PRO_Q8) { if (((uintptr_t) in.act.qs % 16) || ((uintptr_t) in.act.bsums % 16) || !in.act.d) return false; }
    else { if ((!in.x && !(in.planes && in.pl_probs)) || ((uintptr_t) in.x % 16) || (in.mode == PRO_NORM && ((uintptr_t) in.norm_w % 16))) return false; }
    const int64_t nb = b10 ? k/320 : k/256;
    if (in.planes && (in.mode != PRO_NORM || (k + 255)/256 > 16 || !in.pl_probs || in.n_planes < 1 || in.n_planes > 8 || !in.x_out || ((uintptr_t) in.planes % 16) || in.plane_stride % 4 || ((uintptr_t) in.x_out % 16))) return false;
    int ta = -1, tb = -1;
    for (int i = 0; i < n_groups; i++) {
        const mmvq_group & g = groups[i];
        const int ut = st_utype(g.type, k);
        const int ub = st_unit_bytes(ut);
        if (!ub || ((g.type == T_Q8_0 || g.type == T_Q4_0) && in.mode == PRO_Q8)) return false;
        if (ut != ta && ut != tb) { if (ta < 0) ta = ut; else if (tb < 0) tb = ut; else return false; }
        if (b10 && tb >= 0) return false;                              // (one kernel per ten-block format: no mixed launches)
        if ((g.b_gate != nullptr) != (g.b_up != nullptr) || (g.b_gate && (g.epi != EPI_GLU || !g.eid)) || (g.res_eid && (g.epi != EPI_ADD || !g.eid))) return false;
        if (g.eid && (g.estride % 16 || in.mode == PRO_NORM)) return false;
        if (g.x_off && (in.mode == PRO_Q8 || g.x_off % 4)) return false;
        if (g.row_stride != (size_t)(nb*ub) || ((uintptr_t) g.W % 16) || (g.W2 && ((uintptr_t) g.W2 % 16))) return false;
        if (g.epi == EPI_GLU && !g.W2) return false;
        if (g.epi == EPI_ROPE) {
            if (!rope || !rope->table || (g.m & 1)) return false;
            if (rope->p.mode & 2) { const int hd = rope->head_dim; if (hd <= 0 || (hd & (hd - 1)) || rope->p.n_dims != hd || g.m % hd) return false; }
            else if (rope->p.mode != 0) return false;
        }
        if (g.epi == EPI_ADD && !g.res) return false;
        if (g.m < 1) return false;
    }
    // the LDS carve must leave room for a ring of at least two slots (rows whose block count is not a multiple of 16 keep one partial sum per unit: a
    // workgroup with many such rows can fill the LDS with them) — otherwise the register-ring kernels take the launch (the launcher used to abort)
    st_args a; size_t fixed_max; int slot_max, nslots_max, npart_max, fa, fb; double bytes_total;
    (void) st_fill(groups, n_groups, k, in, rope, a, fixed_max, slot_max, nslots_max, npart_max, fa, fb, bytes_total);
    const int64_t S = fixed_max < 163840 ? (163840 - (int64_t) fixed_max)/slot_max : 0;
    if (S < (nslots_max < 2 ? nslots_max : 2)) return false;
    for (int i = 0; i < n_groups; i++) if ((int64_t) 256*groups[i].m >= (1ll << 31)) return false;      // (the kernel deals rows with 32-bit arithmetic: workgroups x row units)
    // experiment knob: launches that stream fewer bytes than this go to the register-ring kernel (is the streamed kernel's envelope a loss on the small launches?)
    static const double min_bytes = getenv("GGML_MI355X_STREAM_MIN_MB") ? atof(getenv("GGML_MI355X_STREAM_MIN_MB"))*1e6 : 0.0;
    if (min_bytes > 0.0 && bytes_total < min_bytes && !in.planes && !b10) {
        for (int i = 0; i < n_groups; i++) if (groups[i].eid || groups[i].epi == EPI_GLU || (groups[i].epi == EPI_ROPE && rope && (rope->p.mode & 2))) return true;
        return false;
    }
    return true;
}

// fills the phase descriptor of one grouped launch; returns the workgroups it uses. fixed: LDS bytes besides the ring; slot: bytes of a ring slot;
// nslots: slots the busiest workgroup streams
static int st_fill(const mmvq_group * groups, int n_groups, int64_t k, const mmvq_input & in, const mmvq_rope * rope, st_args & a,
                   size_t & fixed_max, int & slot_max, int & nslots_max, int & npart_max, int & ta, int & tb, double & bytes_total) {
    a = st_args{};
    const bool b10 = k % 256 != 0;
    const int nb = (int)(b10 ? k/320 : k/256);
    a.n_groups = n_groups; a.k = (int) k; a.nb = nb; a.mode = in.mode; a.eps = in.eps;
    a.nchunk = (int)((k + 255)/256); a.act_stride = !b10 ? ST_ACT_STRIDE : st_utype(groups[0].type, k) == ST_MXFP4_B10 ? ST_ACT_STRIDE_FP4 : ST_ACT_STRIDE_B10;
    a.magic = nb == 1 ? 0u : (uint32_t)((0x100000000ull + nb - 1)/nb);
    a.x = in.x; a.norm_w = in.norm_w;
    a.planes = in.planes; a.n_planes = in.n_planes; a.plane_stride = in.plane_stride; a.x_out = in.x_out;
    a.pl_probs = in.pl_probs; a.pl_ids = in.pl_ids; a.pl_mode = in.pl_mode;
    // (measured, tg128 on one box, `profiles/r04_stream_early_sweep.log`: 0 slots 588.4 tok/s, 1 slot 616.5, 2 slots 613.9, 4 slots 603.3)
    // (not kept: one batch of scalar loads over the whole 928-byte argument block at entry, so that the compiler's ~9 dependent s_loads before the first vector load
    // hit the scalar cache — same box, alternating: 604 / 600 tok/s with the batch, 617 / 617 without: the chain was not missing)
    { static const int early = getenv("GGML_MI355X_STREAM_EARLY") ? atoi(getenv("GGML_MI355X_STREAM_EARLY")) : 1;
      static const int q16 = getenv("GGML_MI355X_STREAM_Q16") ? atoi(getenv("GGML_MI355X_STREAM_Q16")) : 1; 
      static const int touch = getenv("GGML_MI355X_STREAM_XTOUCH") ? atoi(getenv("GGML_MI355X_STREAM_XTOUCH")) : 1; a.early = (early & 0xFF) | (q16 ? 0 : 0x100) | (touch ? 0 : 0x200); }
    if (in.mode == PRO_Q8) { a.a_qs = in.act.qs; a.a_d = in.act.d; a.a_bs = in.act.bsums; }
    if (rope) a.rope = make_fused_rope(*rope);

    // workgroups per group: in proportion to the weight bytes, at least one each, never more than the group has row units
    const int budget = st_cu_count();
    bytes_total = 0; double gbytes[MMVQ_MAX_GROUPS];
    ta = st_utype(groups[0].type, k); tb = ta;
    for (int i = 0; i < n_groups; i++) {
        gbytes[i] = (double) groups[i].m*nb*st_unit_bytes(st_utype(groups[i].type, k))*(groups[i].epi == EPI_GLU ? 2 : 1);
        bytes_total += gbytes[i];
        if (st_utype(groups[i].type, k) != ta) tb = st_utype(groups[i].type, k);
    }
    if (tb < ta) std::swap(ta, tb);
    int blocks = 0; slot_max = 0; fixed_max = 0; nslots_max = 0; npart_max = 0;
    for (int i = 0; i < MMVQ_MAX_GROUPS; i++) a.block_end[i] = INT_MAX;
    int share[MMVQ_MAX_GROUPS], used = 0;
    for (int i = 0; i < n_groups; i++) {
        const mmvq_group & g = groups[i];
        st_group & s = a.g[i];
        s.W = g.W; s.W2 = g.W2; s.dst = g.dst; s.res = g.res; s.res2 = g.res2; s.st16 = g.st16; s.st_idx = g.st_idx; s.st_row_elems = g.st_row_elems;
        s.m = g.m; s.type = st_utype(g.type, k); s.epi = g.epi; s.st_mode = g.st_mode; s.glu_alpha = g.glu_alpha; s.glu_limit = g.glu_limit;
        s.eid = g.eid; s.estride = (long long) g.estride; s.x_off = g.x_off;
        s.b_gate = g.b_gate; s.b_up = g.b_up; s.res_eid = g.res_eid;
        s.ralign = 1; s.neox2 = 0; s.neox_hh = 0;
        if (g.epi == EPI_ROPE) s.ralign = (rope->p.mode & 2) ? rope->head_dim : 2;
        while (((int64_t) s.ralign*nb*st_unit_bytes(s.type)) % 16 != 0) s.ralign *= 2;      // a workgroup's rows start on a 16-byte boundary (LDS-DMA source)
        const int nru = std::max(1, g.m/s.ralign);
        int sh = (int)((double) budget*gbytes[i]/bytes_total);
        sh = sh < 1 ? 1 : (sh > nru ? nru : sh);
        share[i] = sh; used += sh;
    }
    // NEOX groups as two row streams (mmvq_stream.h: st_group::neox2): the smallest number of pairs per workgroup c for which every such group gets EXACTLY m/2/c
    // workgroups and the launch still fits the chip; the other groups share what is left
    static const bool neox2_on = !getenv("GGML_MI355X_NEOX2") || atoi(getenv("GGML_MI355X_NEOX2")) != 0;
    if (neox2_on && rope && (rope->p.mode & 2) && rope->head_dim >= 16) {
        const int hh = rope->head_dim/2;
        for (int c = 4; c < hh; c *= 2) {
            int fixed = 0, others = 0; bool ok = hh % c == 0;
            for (int i = 0; i < n_groups && ok; i++) {
                if (groups[i].epi == EPI_ROPE) { ok = (groups[i].m/2) % c == 0 && ((int64_t) c*nb*st_unit_bytes(a.g[i].type)) % 16 == 0 && ((int64_t) hh*nb*st_unit_bytes(a.g[i].type)) % 16 == 0 && !groups[i].eid; fixed += groups[i].m/2/c; }
                else others++;
            }
            if (!ok || fixed + others > budget) continue;
            double obytes = 0; for (int i = 0; i < n_groups; i++) if (groups[i].epi != EPI_ROPE) obytes += gbytes[i];
            used = 0;
            for (int i = 0; i < n_groups; i++) {
                st_group & s = a.g[i];
                if (groups[i].epi == EPI_ROPE) { s.neox2 = c; s.neox_hh = hh; s.ralign = c; s.W2 = s.W + (size_t) hh*nb*st_unit_bytes(s.type); share[i] = groups[i].m/2/c; }
                else { const int nru = std::max(1, groups[i].m/s.ralign); int sh = (int)((double)(budget - fixed)*gbytes[i]/obytes); share[i] = sh < 1 ? 1 : (sh > nru ? nru : sh); }
                used += share[i];
            }
            break;
        }
    }
    // hand the workgroups rounding left over to the groups with the most bytes per workgroup
    while (used < budget) {
        int best = -1; double bw = 0;
        for (int i = 0; i < n_groups; i++) { const int nru = std::max(1, groups[i].m/a.g[i].ralign); if (!a.g[i].neox2 && share[i] < nru && gbytes[i]/share[i] > bw) { bw = gbytes[i]/share[i]; best = i; } }
        if (best < 0) break;
        share[best]++; used++;
    }
    for (int i = 0; i < n_groups; i++) {
        const mmvq_group & g = groups[i];
        st_group & s = a.g[i];
        const int nru = std::max(1, g.m/s.ralign), sh = share[i];
        const int Rmax = s.neox2 ? s.neox2 : ((nru + sh - 1)/sh)*s.ralign + (g.m - (g.m/s.ralign)*s.ralign);
        const bool row16 = (nb & 15) == 0;
        const int streams = (g.epi == EPI_GLU || s.neox2) ? 2 : 1;
        s.npart_max = (row16 ? Rmax*(nb >> 4) : Rmax*nb)*streams;
        npart_max = std::max(npart_max, s.npart_max);
        const size_t fixed = 2*ST_SYNC_WORDS*4 + (size_t) nb*a.act_stride + (size_t)((nb + 3) & ~3)*4 + 64 + (size_t) s.npart_max*4 + 16;
        fixed_max = std::max(fixed_max, fixed);
        const int pps = (64*st_unit_bytes(s.type) + 1023)/1024;
        slot_max = std::max(slot_max, pps*1024);
        nslots_max = std::max(nslots_max, (int)(((int64_t) Rmax*nb + 63)/64)*streams);
        blocks += sh;
        a.block_end[i] = blocks;
    }
    return blocks;
}

void mul_mat_vec_q_stream(const mmvq_group * groups, int n_groups, int64_t k, const mmvq_input & in, const mmvq_rope * rope, hipStream_t stream,
                          hipEvent_t e0, hipEvent_t e1, const char ** kname) {
    static int ring_env = -1;
    if (ring_env < 0) { const char * r = getenv("GGML_MI355X_STREAM_RING"); ring_env = r ? atoi(r) : 0; }
    st_args a; size_t fixed_max; int slot_max, nslots_max, npart_max, ta, tb; double bytes_total;
    const int blocks = st_fill(groups, n_groups, k, in, rope, a, fixed_max, slot_max, nslots_max, npart_max, ta, tb, bytes_total);
    int S = (int)((163840 - (int64_t) fixed_max)/slot_max);
    if (S > nslots_max) S = nslots_max;
    if (S > ST_MAX_RING) S = ST_MAX_RING;
    if (ring_env > 0 && S > ring_env) S = ring_env;
    if (S < (nslots_max < 2 ? nslots_max : 2)) { fprintf(stderr, "mul_mat_vec_q_stream: no room for a slot ring (k = %lld)\n", (long long) k); abort(); }
    a.S = S;
    const size_t lds = fixed_max + (size_t) S*slot_max;
#ifdef MI_STAMPS
    a.stamps = nullptr;
    if (g_st_stamps && g_st_next < g_st_slots && blocks <= 256) {
        st_stamp_meta & sm = g_st_meta[g_st_next];
        sm.blocks = blocks; sm.k = (int) k; sm.rows = 0; for (int i = 0; i < n_groups; i++) sm.rows += groups[i].m*(groups[i].epi == EPI_GLU ? 2 : 1);
        sm.type_a = ta; sm.type_b = tb; sm.mode = in.mode; sm.glu = groups[0].epi == EPI_GLU; sm.bytes = (long long) bytes_total;
        a.stamps = g_st_stamps + (size_t) g_st_next*256*(ST_NC + 1)*ST_NSTAMP;
        g_st_next++;
    }
#endif
    const bool glu = n_groups == 1 && groups[0].epi == EPI_GLU;
    if (ta == T_Q4_K && tb == T_Q4_K)      { if (glu) st_launch_t<T_Q4_K, T_Q4_K, true>(a, blocks, lds, stream, e0, e1, kname); else st_launch_t<T_Q4_K, T_Q4_K, false>(a, blocks, lds, stream, e0, e1, kname); }
    else if (ta == T_Q5_K && tb == T_Q5_K) { if (glu) st_launch_t<T_Q5_K, T_Q5_K, true>(a, blocks, lds, stream, e0, e1, kname); else st_launch_t<T_Q5_K, T_Q5_K, false>(a, blocks, lds, stream, e0, e1, kname); }
    else if (ta == T_Q6_K && tb == T_Q6_K) { if (glu) st_launch_t<T_Q6_K, T_Q6_K, true>(a, blocks, lds, stream, e0, e1, kname); else st_launch_t<T_Q6_K, T_Q6_K, false>(a, blocks, lds, stream, e0, e1, kname); }
    else if (ta == ST_Q8_0_B10)  { if (glu) st_launch_t<ST_Q8_0_B10, ST_Q8_0_B10, true>(a, blocks, lds, stream, e0, e1, kname); else st_launch_t<ST_Q8_0_B10, ST_Q8_0_B10, false>(a, blocks, lds, stream, e0, e1, kname); }
    else if (ta == ST_MXFP4_B10) { if (glu) st_launch_t<ST_MXFP4_B10, ST_MXFP4_B10, true>(a, blocks, lds, stream, e0, e1, kname); else st_launch_t<ST_MXFP4_B10, ST_MXFP4_B10, false>(a, blocks, lds, stream, e0, e1, kname); }
    else if (ta == T_Q4_0 && tb == T_Q4_0) { if (glu) st_launch_t<T_Q4_0, T_Q4_0, true>(a, blocks, lds, stream, e0, e1, kname); else st_launch_t<T_Q4_0, T_Q4_0, false>(a, blocks, lds, stream, e0, e1, kname); }
    else if (ta == T_Q8_0 && tb == T_Q8_0) { if (glu) st_launch_t<T_Q8_0, T_Q8_0, true>(a, blocks, lds, stream, e0, e1, kname); else st_launch_t<T_Q8_0, T_Q8_0, false>(a, blocks, lds, stream, e0, e1, kname); }
    else if (ta == T_Q8_0 && tb == T_Q4_K) st_launch_t<T_Q8_0, T_Q4_K, false>(a, blocks, lds, stream, e0, e1, kname);
    else if (ta == T_Q8_0 && tb == T_Q6_K) st_launch_t<T_Q8_0, T_Q6_K, false>(a, blocks, lds, stream, e0, e1, kname);
    else if (ta == T_Q4_K && tb == T_Q5_K) st_launch_t<T_Q4_K, T_Q5_K, false>(a, blocks, lds, stream, e0, e1, kname);
    else if (ta == T_Q4_K && tb == T_Q6_K) st_launch_t<T_Q4_K, T_Q6_K, false>(a, blocks, lds, stream, e0, e1, kname);
    else if (ta == T_Q5_K && tb == T_Q6_K) st_launch_t<T_Q5_K, T_Q6_K, false>(a, blocks, lds, stream, e0, e1, kname);
    else { fprintf(stderr, "mul_mat_vec_q_stream: type pair (%d, %d) has no kernel\n", ta, tb); abort(); }
}

} // namespace mi355x
