// mmvq_fused.hip — host side of the persistent grouped mat-vec (device code: mmvq_fused.h, kernels: mmvq_fused_<type>.hip):
// shares the workgroups among the groups, fills the argument block, picks the kernel family.
#include "mmvq_fused.h"

#include <limits.h>

namespace mi355x {

#ifdef MI_STAMPS
// debug build only (-DMI_STAMPS): every grouped mat-vec launch gets the next slot of a device buffer; slots are baked into captured graphs
static unsigned long long * g_stamp_buf = nullptr;
static int g_stamp_slots = 0, g_stamp_next = 0;
constexpr int STAMP_MAX_WG = 1024;
struct stamp_meta { int blocks, k, rows, type_a, type_b, mode, glu; long long bytes; };
static stamp_meta g_stamp_meta[4096];
extern "C" int mi355x_stamps_enable(int n_slots) {
    if (g_stamp_buf) { (void) hipFree(g_stamp_buf); g_stamp_buf = nullptr; }
    g_stamp_slots = n_slots > 4096 ? 4096 : n_slots; g_stamp_next = 0;
    if (g_stamp_slots <= 0) return 0;
    if (hipMalloc(&g_stamp_buf, (size_t) g_stamp_slots*STAMP_MAX_WG*MI_STAMP_N*8) != hipSuccess) return -1;
    (void) hipMemset(g_stamp_buf, 0, (size_t) g_stamp_slots*STAMP_MAX_WG*MI_STAMP_N*8);
    return 0;
}
extern "C" int mi355x_stamps_used(void) { return g_stamp_next; }
extern "C" int mi355x_stamps_read(int slot, unsigned long long * out, int * meta, long long * bytes) {
    if (!g_stamp_buf || slot < 0 || slot >= g_stamp_slots) return -1;
    const stamp_meta & m = g_stamp_meta[slot];
    (void) hipMemcpy(out, g_stamp_buf + (size_t) slot*STAMP_MAX_WG*MI_STAMP_N, (size_t) m.blocks*MI_STAMP_N*8, hipMemcpyDeviceToHost);
    meta[0] = m.blocks; meta[1] = m.k; meta[2] = m.rows; meta[3] = m.type_a; meta[4] = m.type_b; meta[5] = m.mode; meta[6] = m.glu;
    *bytes = m.bytes;
    return 0;
}
#endif

static size_t pad256h(size_t x) { return (x + 255) & ~(size_t) 255; }

static size_t act_image_bytes(int64_t k, int act_kind) {
    const int64_t nd = act_kind == T_Q8_0 ? k/32 : k/256, nbs = act_kind == T_Q8_0 ? k/32 : k/16;
    return pad256h(k) + pad256h(nd*4) + ((nbs*2 + 15) & ~15);
}

// the activation must be the n = 1 image act_q8_carve lays out: qs | pad | d | pad | bsums, contiguous
bool mul_mat_vec_q_fused_supported(int64_t k, int act_kind) {
    return k % (act_kind == T_Q8_0 ? 32 : 256) == 0 && act_image_bytes(k, act_kind) <= 4*512*16;
}
bool mul_mat_vec_q_fused_prologue_supported(int64_t k, int act_kind) { return (k % 256 == 0 || (act_kind == T_Q8_0 && k % 32 == 0)) && k <= 16*1024; }

// per host thread: one backend (stream) is driven by one thread at a time, different backends concurrently from different threads
// (tests/test-thread-safety.cpp)
static thread_local struct { mmvq_launch_hook pre = nullptr, post = nullptr; void * ctx = nullptr; } g_hook;
hipEvent_t mi355x_fused_ev0 = nullptr, mi355x_fused_ev1 = nullptr;
const char * mi355x_fused_last_kernel = "";
const char * mul_mat_vec_q_fused_last_kernel(void) { return mi355x_fused_last_kernel; }
void mul_mat_vec_q_fused_set_launch_events(hipEvent_t e0, hipEvent_t e1) { mi355x_fused_ev0 = e0; mi355x_fused_ev1 = e1; }
void mul_mat_vec_q_fused_set_hooks(mmvq_launch_hook pre, mmvq_launch_hook post, void * ctx) { g_hook.pre = pre; g_hook.post = post; g_hook.ctx = ctx; }

static int device_cu_count() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    return n_cu;
}

// how the persistent workgroups (one per CU) are shared among the groups of a launch: in proportion to their rows, at least one each,
// never more than one unit (row pair; row of the dual GLU stream) per wave. Returns the grid size; block_end[i] = cumulative counts.
static int share_groups(const mmvq_group * groups, int n_groups, int fw, int budget, int * block_end) {
    int blocks = 0;
    if (fw == 16) {
        // 16-wave workgroups own units wave*nwg + wg: any count up to the group's units is balanced. Shares in proportion to units x a
        // per-format weight (a Q6_K pair takes ~2x as long to queue and stream as a Q4_K pair), rounded down, the budget never exceeded
        double tot = 0.0;
        for (int i = 0; i < n_groups; i++) tot += (double)((groups[i].m + 1)/2)*(groups[i].type == T_Q6_K ? 2.0 : 1.0);
        for (int i = 0; i < n_groups; i++) {
            const int64_t units = (groups[i].m + 1)/2;
            int share = (int)((double) budget*(double) units*(groups[i].type == T_Q6_K ? 2.0 : 1.0)/tot);
            share = share < 1 ? 1 : (share > units ? (int) units : share);
            blocks += share;
            block_end[i] = blocks;
        }
        return blocks;
    }
    int64_t rows_total = 0;
    for (int i = 0; i < n_groups; i++) rows_total += (int64_t) groups[i].m*(groups[i].epi == EPI_GLU ? 2 : 1);
    for (int i = 0; i < n_groups; i++) {
        const int max_wg = groups[i].epi == EPI_GLU ? (int)((groups[i].m + fw - 1)/fw) : (int)(((groups[i].m + 1)/2 + fw - 1)/fw);   // units: rows (GLU) or row pairs
        int share = (int)(((int64_t) budget*groups[i].m*(groups[i].epi == EPI_GLU ? 2 : 1))/rows_total);   // rounded down: the grid never exceeds the budget (one workgroup per CU)
        share = share < 1 ? 1 : (share > max_wg ? max_wg : share);
        blocks += share;
        block_end[i] = blocks;
    }
    return blocks;
}
int mul_mat_vec_q_fused_share(const mmvq_group * groups, int n_groups, int fw, int * block_end) {
    return share_groups(groups, n_groups, fw, device_cu_count(), block_end);
}

// ---- the rotation table of one token: tab[2*ip] = cos(theta_ip)*mscale, tab[2*ip + 1] = sin(theta_ip)*mscale, ip < n_dims/2, for the position
//      pos[0] — elem.hip's k_rope / rope_pair formulas (YaRN ramp, frequency factors), one lane per pair index ----
__global__ void __launch_bounds__(256) k_rope_table(const fused_rope r, float * tab) {
    const int ip = blockIdx.x*256 + threadIdx.x;
    if (ip >= (r.n_dims >> 1)) return;
    const int pos = r.pos[0];
    const float theta_base = (float) pos*powf(r.theta_scale, (float) ip);
    const float theta_extrap = theta_base/(r.ff ? r.ff[ip] : 1.0f);
    float theta_interp = r.freq_scale*theta_extrap, theta = theta_interp, mscale = r.attn_factor;
    if (r.ext_factor != 0.0f) {
        const float y = ((float) ip - r.corr_lo)/fmaxf(0.001f, r.corr_hi - r.corr_lo);
        const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, y)))*r.ext_factor;
        theta = theta_interp*(1.0f - ramp_mix) + theta_extrap*ramp_mix;
        mscale *= 1.0f + 0.1f*logf(1.0f/r.freq_scale);
    }
    tab[2*ip] = cosf(theta)*mscale; tab[2*ip + 1] = sinf(theta)*mscale;
}
void mul_mat_vec_q_fused_rope_table(const mmvq_rope & rope, float * table, hipStream_t stream) {
    const fused_rope r = make_fused_rope(rope);
    const int np = rope.p.n_dims/2;
    hipLaunchKernelGGL(k_rope_table, dim3((unsigned)((np + 255)/256)), dim3(256), 0, stream, r, table);
}

bool mul_mat_vec_q_fused_fin_supported(int64_t m, int64_t k_in) {
    // chunks are 256 rows; a workgroup's contiguous run (8 waves x rows per wave, one workgroup per CU) may touch at most 7 of them
    const int64_t rpw = (m + (int64_t) device_cu_count()*8 - 1)/((int64_t) device_cu_count()*8);
    return m % 256 == 0 && m >= 256 && 8*rpw <= 1024 && k_in % 256 == 0;
}

static fused_launch fused_prepare(const mmvq_group * groups, int n_groups, int64_t k, const mmvq_input & in, const mmvq_rope * rope, const mmvq_fin * fin) {
    fused_launch L = {};
    fused_mmvq_args & a = L.a;
    if (fin && fin->kind && n_groups == 1 && groups[0].epi == EPI_GLU && !groups[0].eid && mul_mat_vec_q_fused_fin_supported(groups[0].m, k)) a.fin = *fin;
    a.n_groups = n_groups; a.k = (int) k; a.act_kind = in.act_kind;
    // share the persistent workgroups among the groups in proportion to their rows (never more than one row pair per wave)
    static int wpc = 0, glu_wpc = 1;   // measured (tools/stamp_timeline.py): the second workgroup on a CU runs its prologue ~2x slower
    const int n_cu = device_cu_count();
    if (wpc == 0) {
        wpc = 1;
        if (const char * e = getenv("GGML_MI355X_MMVQ_WPC")) wpc = atoi(e) > 0 ? atoi(e) : 1;
        if (const char * e = getenv("GGML_MI355X_GLU_WPC")) glu_wpc = atoi(e) > 0 ? atoi(e) : 1;
    }
    int64_t rows_total = 0;
    for (int i = 0; i < n_groups; i++) rows_total += (int64_t) groups[i].m*(groups[i].epi == EPI_GLU ? 2 : 1);
    // 16 waves per workgroup: more than one row pair per wave at 8 waves x CUs, at most one at 16 (the in-prologue-norm launches only)
    static int fw16_env = -1;
    if (fw16_env < 0) { const char * e = getenv("GGML_MI355X_MMVQ_FW16"); fw16_env = e ? atoi(e) : 1; }
    const int FW = (fw16_env && groups[0].epi != EPI_GLU && in.mode == PRO_NORM && k <= 4096 && k % 256 == 0 &&
                    (rows_total + 1)/2 > (int64_t) n_cu*8 &&
                    ((rows_total + 1)/2 <= (int64_t) n_cu*16 + 64 || (rows_total + 1)/2 >= (int64_t) n_cu*64)) ? 16 : 8;   // or a long stream (lm_head: 101 -> 96 us)
    L.fw = FW;
    const int budget = n_cu*(groups[0].epi == EPI_GLU ? glu_wpc : wpc);   // the dual (GLU) kernels need > 128 VGPRs: one workgroup per CU
    for (int i = 0; i < MMVQ_MAX_GROUPS; i++) {
        a.block_end[i] = INT_MAX; a.x_off[i] = 0; a.gtype[i] = groups[0].type; a.gm[i] = 1; a.gW[i] = groups[0].W; a.gW2[i] = groups[0].W2;
        a.geid[i] = nullptr; a.kidx[i] = nullptr; a.grow_stride[i] = 0; a.gestride[i] = 0;
    }
    for (int i = 0; i < n_groups; i++) {
        a.g[i] = groups[i];
        a.x_off[i] = groups[i].x_off; a.gtype[i] = groups[i].type; a.gm[i] = groups[i].m; a.gW[i] = groups[i].W; a.gW2[i] = groups[i].W2;
        if (groups[i].epi == EPI_ROPE && rope && (rope->p.mode & 2)) {
            // NEOX rotation: a unit of this group is the partner rows i and i + n_dims/2 of one head (the caller checked n_dims == head size = 2^(h+1))
            int h = 0; while ((2 << h) < rope->head_dim) h++;
            if ((2 << h) != rope->head_dim || rope->p.n_dims != rope->head_dim || groups[i].m % rope->head_dim != 0) { fprintf(stderr, "mul_mat_vec_q_fused: NEOX rope needs n_dims == head size == a power of two\n"); abort(); }
            a.gtype[i] |= (h + 1) << 16;
        }
        a.geid[i] = groups[i].eid; a.kidx[i] = groups[i].st_mode == 1 ? groups[i].st_idx : nullptr;
        if (groups[i].row_stride > 0xFFFFFFFFull || groups[i].estride > 0xFFFFFFFFull) { fprintf(stderr, "mul_mat_vec_q_fused: row / expert stride beyond 4 GiB\n"); abort(); }
        a.grow_stride[i] = (uint32_t) groups[i].row_stride; a.gestride[i] = (uint32_t) groups[i].estride;
    }
    int be_[MMVQ_MAX_GROUPS];
    const int blocks = share_groups(groups, n_groups, FW, budget, be_);
    for (int i = 0; i < n_groups; i++) a.block_end[i] = be_[i];
    const int64_t nd = in.act_kind == T_Q8_0 ? k/32 : k/256;
    a.off_d = (int) pad256h(k);
    a.off_bs = (int)(pad256h(k) + pad256h(nd*4));
    const size_t bytes = act_image_bytes(k, in.act_kind);
    a.act_chunks = (int)(bytes/16);
    if (in.mode == PRO_Q8) {
        a.act = (const char *) in.act.qs;
        if ((const char *) in.act.d - (const char *) in.act.qs != a.off_d || (const char *) in.act.bsums - (const char *) in.act.qs != a.off_bs) {
            fprintf(stderr, "mul_mat_vec_q_fused: activation is not a contiguous n = 1 image\n"); abort();
        }
    } else {
        a.x = in.x; a.norm_w = in.norm_w; a.eps = in.eps;
    }
    a.pos = nullptr;
    if (rope) {
        a.rope = make_fused_rope(*rope);
        a.pos = rope->pos;
        if (!rope->table) { fprintf(stderr, "mul_mat_vec_q_fused: a launch with a rotation needs the token's table (mul_mat_vec_q_fused_rope_table)\n"); abort(); }
    }
    size_t img_max = bytes;                   // groups of another activation format build a different image (PRO_QUANT / PRO_NORM only)
    for (int i = 0; i < n_groups; i++) {
        const int kd = act_kind_for(groups[i].type);
        if (kd != in.act_kind) {
            if (in.mode == PRO_Q8) { fprintf(stderr, "mul_mat_vec_q_fused: groups of two activation formats cannot share a copied image\n"); abort(); }
            img_max = std::max(img_max, act_image_bytes(k, kd));
        }
    }
    size_t lds = img_max + 64 + 64;           // + FW floats for the RMS reduction + the finaliser's chunk list
    int ta = groups[0].type, tb = groups[0].type;
    for (int i = 1; i < n_groups; i++) if (groups[i].type != ta) tb = groups[i].type;
    if (tb < ta) { const int t = ta; ta = tb; tb = t; }
    const bool glu = groups[0].epi == EPI_GLU;
    const int mode = in.mode;
#ifdef MI_STAMPS
    a.stamps = nullptr;
    if (g_stamp_buf && g_stamp_next < g_stamp_slots && blocks <= STAMP_MAX_WG) {
        stamp_meta & sm = g_stamp_meta[g_stamp_next];
        sm.blocks = blocks; sm.k = (int) k; sm.rows = (int) rows_total; sm.type_a = ta; sm.type_b = tb; sm.mode = mode; sm.glu = glu;
        sm.bytes = 0;
        for (int i = 0; i < n_groups; i++) sm.bytes += (long long) groups[i].m*groups[i].row_stride*(groups[i].epi == EPI_GLU ? 2 : 1);
        a.stamps = g_stamp_buf + (size_t) g_stamp_next*STAMP_MAX_WG*MI_STAMP_N;
        g_stamp_next++;
    }
#endif
    const int na = mode == PRO_Q8 ? (a.act_chunks <= 512 ? 1 : (a.act_chunks <= 1024 ? 2 : 4)) : (FW == 16 ? 1 : (k <= 4096 ? 2 : 8));
    // prefetch depth: measured on Llama-3-8B Q4_K_M tg128 (profiles/r01_g_*): D = 2 everywhere 503 tok/s, D = 4 (3 for the dual GLU
    // stream) everywhere 482-486 — a CU's request queue is finite and a wave that cannot queue a load cannot run its share of the
    // prologue either. Only long single-tensor streams (>= 16 steps per wave: the lm_head, 31 row pairs per wave) take the deep ring;
    // GGML_MI355X_MMVQ_DEPTH=2|4 forces one for experiments (the GLU kernels exist with D = 4 only).
    static int depth_env = -1;
    if (depth_env < 0) { const char * e = getenv("GGML_MI355X_MMVQ_DEPTH"); depth_env = e ? atoi(e) : 0; }
    int64_t max_steps = 0;
    for (int i = 0; i < n_groups; i++) {
        const int nwg_i = a.block_end[i] - (i ? a.block_end[i - 1] : 0);
        const int64_t pairs = (groups[i].m + 1)/2, per_wave = (pairs + (int64_t) nwg_i*FW - 1)/((int64_t) nwg_i*FW);
        const int kd = act_kind_for(groups[i].type);
        const int64_t nblk = k/(kd == T_Q8_0 ? 32 : 256);
        const int64_t it = kd == T_Q8_0 ? (nblk + 63)/64 : (nblk + 7)/8;      // <= the steps per row pair of every type
        if (per_wave*it > max_steps) max_steps = per_wave*it;
    }
    const bool deep = depth_env ? depth_env > 2 : max_steps >= 16;
    L.ext = false;      // does any group need the extended epilogue
    for (int i = 0; i < n_groups; i++)
        if ((groups[i].epi == EPI_ROPE && (groups[i].res || (rope && (rope->p.mode & 2)))) || groups[i].res2 || groups[i].res_eid) L.ext = true;
    L.blocks = blocks; L.lds = lds; L.ta = ta; L.tb = tb; L.glu = glu; L.mode = mode; L.na = na; L.deep = deep; L.k = k;
    for (int i = 0; i < n_groups; i++) L.wbytes += (uint64_t) groups[i].m*groups[i].row_stride*(groups[i].epi == EPI_GLU ? 2 : 1);
    return L;
}

static void fused_launch_kernel(const fused_launch & L, hipStream_t stream) {
    const int ta = L.ta, tb = L.tb;
    if (ta == tb) {
        switch (ta) {
            case T_Q4_K:  launch_fused_q4_K(L, stream); return;
            case T_Q5_K:  launch_fused_q5_K(L, stream); return;
            case T_Q6_K:  launch_fused_q6_K(L, stream); return;
            case T_Q8_0:  launch_fused_q8_0(L, stream); return;
            case T_Q4_0:  launch_fused_q4_0(L, stream); return;
            case T_MXFP4: launch_fused_mxfp4(L, stream); return;
            default: break;
        }
    } else if (!L.glu) {
        if (ta == T_Q4_K && tb == T_Q5_K) { launch_fused_q4_K_q5_K(L, stream); return; }
        if (ta == T_Q4_K && tb == T_Q6_K) { launch_fused_q4_K_q6_K(L, stream); return; }
        if (ta == T_Q5_K && tb == T_Q6_K) { launch_fused_q5_K_q6_K(L, stream); return; }
        if (ta == T_Q8_0 && tb == T_Q4_K && L.mode != PRO_Q8) { launch_fused_q8_0_q4_K(L, stream); return; }
    }
    fprintf(stderr, "mul_mat_vec_q_fused: type pair (%d, %d) has no kernel (check mul_mat_vec_q_fused_can_group)\n", ta, tb);
    abort();
}

int mul_mat_vec_q_fused_pending(uint64_t * wbytes) { if (wbytes) *wbytes = 0; return 0; }     // nothing is ever held back (the round-1 chained launch is gone)
void mul_mat_vec_q_fused_flush(hipStream_t) { }

void mul_mat_vec_q_fused(const mmvq_group * groups, int n_groups, int64_t k, const mmvq_input & in_, const mmvq_rope * rope, hipStream_t stream,
                         const mmvq_fin * fin) {
    mmvq_input in = in_;
    if (in.planes && !mul_mat_vec_q_stream_takes(groups, n_groups, k, in, rope)) {
        // a pending MoE combine and a consumer that is not the streamed kernel: evaluate it first, then the vector exists like any other
        moe_combine(in.pl_probs, in.pl_ids, in.n_planes, in.pl_mode, in.planes, (size_t) in.plane_stride*4, k, in.x, in.x_out, stream);
        in.x = in.x_out; in.planes = nullptr; in.n_planes = 0; in.x_out = nullptr; in.pl_probs = nullptr; in.pl_ids = nullptr;
    }
    if (mul_mat_vec_q_stream_takes(groups, n_groups, k, in, rope)) {      // the streamed kernel (mmvq_stream.h); callers do not pass it a `fin`
        uint64_t wbytes = 0;
        for (int i = 0; i < n_groups; i++) wbytes += (uint64_t) groups[i].m*groups[i].row_stride*(groups[i].epi == EPI_GLU ? 2 : 1);
        if (g_hook.pre) g_hook.pre(g_hook.ctx, groups[0].type, wbytes, 1, k);
        hipEvent_t e0 = mi355x_fused_ev0, e1 = mi355x_fused_ev1;
        mi355x_fused_ev0 = nullptr; mi355x_fused_ev1 = nullptr;
        mul_mat_vec_q_stream(groups, n_groups, k, in, rope, stream, e0, e1, &mi355x_fused_last_kernel);
        if (g_hook.post) g_hook.post(g_hook.ctx, groups[0].type, wbytes, 1, k);
        return;
    }
    const fused_launch L = fused_prepare(groups, n_groups, k, in, rope, fin);
    if (g_hook.pre) g_hook.pre(g_hook.ctx, L.ta, L.wbytes, 1, L.k);
    fused_launch_kernel(L, stream);
    static const int dup = getenv("GGML_MI355X_DEBUG_DUP_LAUNCH") ? atoi(getenv("GGML_MI355X_DEBUG_DUP_LAUNCH")) : 0;     // timing experiments only (results are wrong where dst aliases the residual)
    if (dup) {
        fused_launch L2 = fused_prepare(groups, n_groups, k, in, rope, fin);      // (its own stamp slot in the -DMI_STAMPS build)
        fused_launch_kernel(L2, stream);
    }
    if (g_hook.post) g_hook.post(g_hook.ctx, L.ta, L.wbytes, 1, L.k);
}

// which weight types may share one grouped launch (the mixtures llama_tensor_get_type produces, src/llama-quant.cpp:178-434)
bool mul_mat_vec_q_fused_can_group(int type_a, int type_b) {
    if (type_a == type_b) return true;
    const int lo = type_a < type_b ? type_a : type_b, hi = type_a < type_b ? type_b : type_a;
    return (lo == T_Q4_K && (hi == T_Q5_K || hi == T_Q6_K)) || (lo == T_Q5_K && hi == T_Q6_K);
}
// ... and which pairs of DIFFERENT activation formats may, when every workgroup quantizes the activation itself (PRO_QUANT / PRO_NORM)
bool mul_mat_vec_q_fused_can_group_mixed(int type_a, int type_b) {
    const int lo = type_a < type_b ? type_a : type_b, hi = type_a < type_b ? type_b : type_a;
    return lo == T_Q8_0 && hi == T_Q4_K;
}

} // namespace mi355x
