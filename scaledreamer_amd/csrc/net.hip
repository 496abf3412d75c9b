// net.hip — the layer schedules of the frozen diffusion prior, enqueued from C++ on the caller's stream:
//   asd_unet_*     SD-2.1 / MVDream UNet eps-prediction   (stable_diffusion_asd_guidance.py:319-331 forward_unet -> diffusers
//                  UNet2DConditionModel; in-tree statement of the arithmetic: extern/mvdream/ldm/modules/diffusionmodules/
//                  openaimodel.py:771-808 UNetModel.forward, :1175-1213 MultiViewUNetModel.forward, ResBlock :252-275,
//                  attention.py:246-275 BasicTransformerBlock, :320-354 SpatialTransformer(3D))
//   asd_vae_enc_*  VAE encoder forward + input gradient    (stable_diffusion_asd_guidance.py:171-178 encode_images; Encoder
//                  diffusionmodules/model.py:452-543, ResnetBlock :88-146, Downsample :66-85, AttnBlock :152-227, quant_conv
//                  models/autoencoder.py:32,81-85)
// A network is a handle created once from its hyper-parameters.  It publishes the table of PACKED weights it reads (name, rows,
// cols: conv -> [Cout][ky][kx][Cin], q|k fused, every time-embedding projection fused into one matrix, ...), the caller binds
// device pointers to that table and owns the workspace; a forward is a single chain of kernel launches with no host
// synchronisation, so it can be captured into a HIP graph by any host.  No kernels live here: the schedules call the leaf
// entry points of gemm.hip / nn_ops.hip / attention.hip.
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "asd_common.h"

namespace {

typedef _Float16 half_t;

struct WSpec { std::string name; int rows, cols; };

struct Bump {               // linear allocator over the caller's workspace; base == nullptr: dry run (sizes only)
    char* base = nullptr;
    size_t off = 0, cap = 0;
    void* take(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
    half_t* halfs(size_t n) { return (half_t*)take(n * 2); }
    float* floats(size_t n) { return (float*)take(n * 4); }
};

struct Run {                // state of one pass over a schedule
    Bump mem;
    hipStream_t stream = nullptr;
    bool dry = true;        // no launches
    bool tune = false;      // time un-tuned GEMM shapes before launching them
    float* scratch = nullptr;   // split-K slabs
    size_t scratch_bytes = 0, scratch_need = 0;
    const void* zero_page = nullptr;
    int status = ASD_OK;
    // GroupNorm-statistics records written by producers' epilogues (asd_gemm_args.gn_partials): their count depends on the GEMM
    // plan, so a pass that must reproduce another pass's allocation sequence (the VAE backward replaying its forward) replays the
    // logged counts instead of asking the plan table again
    std::vector<int>* rec_log = nullptr;
    const std::vector<int>* rec_replay = nullptr;
    size_t rec_pos = 0;
};

struct Act {                // an activation and, when its producer left them, its GroupNorm statistics records
    const half_t* p = nullptr;
    const float* rec = nullptr;
    int nrec = 0;           // records per batch element (0: none — the consumer runs its own statistics pass)
    // GroupNorm(+SiLU) of p already applied by the producer's split-K reduction (asd_gemm_args.gn_apply) with these parameters
    const half_t* applied = nullptr; const half_t* applied_gamma = nullptr; const half_t* applied_beta = nullptr;
    float applied_eps = 0.f; int applied_silu = 0;
};

struct GnNext {             // the GroupNorm that consumes a layer's output, when the schedule knows it
    const half_t* gamma = nullptr; const half_t* beta = nullptr; float eps = 0.f; int silu = 0;
};

struct Net {
    std::vector<WSpec> specs;
    std::map<std::string, int> index;
    std::vector<const void*> ptr;
    void* zero_page = nullptr;
    bool bound = false;
    int add(const std::string& name, int rows, int cols) {
        index[name] = (int)specs.size();
        specs.push_back(WSpec{name, rows, cols});
        ptr.push_back(nullptr);
        return (int)specs.size() - 1;
    }
    const half_t* w(const std::string& name) const {
        auto it = index.find(name);
        return it == index.end() ? nullptr : (const half_t*)ptr[it->second];
    }
    bool has(const std::string& name) const { return index.count(name) != 0; }
};

int pad32(int c) { return (c + 31) / 32 * 32; }

// ---- leaf launches ---------------------------------------------------------------------------------------------------------
struct GemmOpt {
    const void* bias = nullptr; const void* row_bias = nullptr; int rows_per_group = 1, ld_row_bias = 0;
    const void* residual = nullptr; int ldr = 0; int act = 0; int out_f32 = 0;
    int gn_rows = 0;        // > 0: the output feeds a GroupNorm over all N channels with this many rows per batch element
    Act* out = nullptr;     // receives the records' location when gn_rows > 0
    // backward form (the output is the gradient dy reaching a GroupNorm with input gn_x): records of {sum g, sum g * xhat}
    const half_t* gn_x = nullptr; const float* gn_fstats = nullptr; const half_t* gn_gamma = nullptr; const half_t* gn_beta = nullptr;
    int gn_silu = 0;
    bool gn_bwd_form = false;   // set with gn_x (dry passes carry null pointers: the record count must still be the backward form's)
    // LayerNorm folded into this GEMM (asd_gemm_args.ln_mode): 1 = the rows of A are normalised (statistics reduced in the main loop,
    // optionally exported to ln_stats), 2 = the rows of the W operand are (statistics read from ln_stats)
    int ln_mode = 0; const float* ln_sc = nullptr; float* ln_stats = nullptr;
    // the GroupNorm that consumes this output (with gn_rows / out): a split-K launch applies it in its reduction kernel and leaves the
    // normalised tensor in out->applied (asd_gemm_args.gn_apply); otherwise the records path above
    const GnNext* gn_next = nullptr;
};

void set_gn_bwd(asd_gemm_args& g, const GemmOpt& o) {
    g.gn_bwd_x = o.gn_x; g.gn_bwd_fstats = o.gn_fstats; g.gn_bwd_gamma = o.gn_gamma; g.gn_bwd_beta = o.gn_beta;
    g.gn_eps = 1e-6f; g.gn_silu = o.gn_silu;
}

void launch_gemm(Run& r, asd_gemm_args& g, Act* gn_out = nullptr, bool gn_bwd_form = false, const GnNext* gn_next = nullptr) {
    g.zero_page = r.zero_page;
    g.split_k = 0;          // auto: tuned plan of this shape (asd_gemm_plan_*), else cost model
    g.tile_cfg = 0;
    if (r.tune && !r.dry) {
        int32_t t, s;
        if (asd_gemm_plan_get(&g, &t, &s) != ASD_OK) {       // 1 = not tuned yet
            if (asd_gemm_tune(&g, r.scratch, (int64_t)r.scratch_bytes, r.stream) != ASD_OK) { r.status = ASD_ERR_LAUNCH; return; }
        }
    }
    const size_t need = (size_t)asd_gemm_workspace_bytes(&g);
    if (need > r.scratch_need) r.scratch_need = need;
    static const bool gn_epilogue = !(getenv("ASD_GN_EPILOGUE") && getenv("ASD_GN_EPILOGUE")[0] == '0');    // A/B switch (tools)
    // The backward form of the records (the two reductions of the GroupNorm input gradient from the dgrad launch's epilogue) is OFF by
    // default since round 3: on the ping-pong kernel the epilogue's silu'(z) over the whole tile is not hidden behind another block's
    // main loop — the VAE's eleven dgrad launches grew by 22-53 us each (0.39 ms per step) for 0.33 ms of statistics passes saved; same-box
    // A/B 15.58 vs 15.64 ms per step (gpurun_out/gnb).  ASD_GN_BWD_EPILOGUE=1 switches it back on.
    static const bool gn_bwd_epilogue = getenv("ASD_GN_BWD_EPILOGUE") && getenv("ASD_GN_BWD_EPILOGUE")[0] == '1';
    if (gn_bwd_form && !gn_bwd_epilogue) { gn_out = nullptr; g.gn_bwd_x = nullptr; g.gn_bwd_fstats = nullptr; g.gn_bwd_gamma = nullptr; g.gn_bwd_beta = nullptr; }
    static const bool gn_fused_apply = !(getenv("ASD_GN_FUSED_APPLY") && getenv("ASD_GN_FUSED_APPLY")[0] == '0');          // A/B switch (tools)
    if (gn_fused_apply && gn_epilogue && gn_next && gn_next->gamma && gn_out && !gn_bwd_form && g.gn_rows > 0 && g.N % 32 == 0 && !r.rec_replay && !r.rec_log) {
        // GroupNorm applied by the producer: decided by shape and plan only (dry passes carry null pointers)
        asd_gemm_args q = g;
        q.gn_cg = g.N / 32; q.gn_apply = 1;
        const bool applies = asd_gemm_gn_applies(&q) != 0;
        // while tuning the plan of this shape may still change between the sizing pass and the launch: always reserve the buffer
        half_t* y = (applies || r.tune) ? r.mem.halfs((size_t)g.M * g.N) : nullptr;
        if (applies) {
            g.gn_cg = g.N / 32; g.gn_apply = 1; g.gn_apply_y = y;
            g.gn_apply_gamma = gn_next->gamma; g.gn_apply_beta = gn_next->beta; g.gn_apply_eps = gn_next->eps; g.gn_apply_silu = gn_next->silu;
            gn_out->applied = y; gn_out->applied_gamma = gn_next->gamma; gn_out->applied_beta = gn_next->beta;
            gn_out->applied_eps = gn_next->eps; gn_out->applied_silu = gn_next->silu;
            gn_out = nullptr;           // no records: nothing else reads this tensor's statistics
        }
    }
    if (gn_epilogue && gn_out && g.gn_rows > 0 && g.N % 32 == 0) {     // statistics records of the output, produced in the epilogue when the plan allows
        g.gn_cg = g.N / 32;
        const int batch = g.M / g.gn_rows;
        int nrec;
        if (r.rec_replay) nrec = r.rec_pos < r.rec_replay->size() ? (*r.rec_replay)[r.rec_pos++] : 0;
        else {
            asd_gemm_args q = g;             // the answer depends on shape, plan and form only (dry passes carry null pointers)
            q.gn_bwd_x = nullptr;
            nrec = asd_gemm_gn_records(&q);
            if (gn_bwd_form && q.split_k == 0) {        // the backward form has no split-K variant (splitk_epilogue_gn_kernel is forward only)
                int32_t t = 0, sk = 1;
                asd_gemm_plan_get(&q, &t, &sk);
                if (sk > 1) nrec = 0;
            }
        }
        // while tuning, the plan (and with it the record count) of this shape may still change between the sizing pass and the
        // launch: reserve the upper bound (64 x 64 tiles)
        const int reserve = r.tune ? (g.gn_rows / 64 + 1) * (g.N / 64 + 1) : nrec;
        if (r.rec_log) r.rec_log->push_back(nrec);
        float* buf = (nrec > 0 || r.tune) ? r.mem.floats((size_t)batch * (reserve > nrec ? reserve : nrec) * 64) : nullptr;
        if (nrec > 0) { g.gn_partials = buf; gn_out->rec = buf; gn_out->nrec = nrec; }
    }
    if (r.dry || r.status != ASD_OK) return;
    if (need > r.scratch_bytes) { asd_set_error("network workspace too small for a split-K GEMM"); r.status = ASD_ERR_ARG; return; }
    g.workspace = r.scratch;
    const int st = asd_gemm_f16(&g, r.stream);
    if (st != ASD_OK) r.status = st;
}

// C[M,N] = act(A[M,K] W[N,K]^T + bias + row_bias) + residual
void gemm(Run& r, const void* A, int M, int lda, const void* W, int N, int K, int ldw, void* C, int ldc, const GemmOpt& o = GemmOpt()) {
    asd_gemm_args g = asd_gemm_args{};
    g.A = A; g.W = W; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldw = ldw; g.ldc = ldc;
    g.bias = o.bias; g.row_bias = o.row_bias; g.rows_per_group = o.row_bias ? o.rows_per_group : 1; g.ld_row_bias = o.ld_row_bias;
    g.residual = o.residual; g.ldr = o.ldr; g.act = o.act; g.out_f32 = o.out_f32;
    g.gn_rows = o.gn_rows;
    g.ln_mode = o.ln_mode; g.ln_sc = o.ln_sc; g.ln_stats = o.ln_stats; g.ln_eps = 1e-5f;
    if (o.gn_x) set_gn_bwd(g, o);
    if (o.out) { *o.out = Act{}; o.out->p = (const half_t*)C; }
    launch_gemm(r, g, o.out, o.gn_bwd_form, o.gn_next);
}

// 3x3 convolution on NHWC [B,Hin,Win,Cin] with packed weights [Cout, 9*Cin]; upsample: 0 plain, 1 nearest-2x fused, 2 transposed stride-2,
// 3 nearest-2x fused in its parity form (four 2x2 convolutions over the low-resolution image, weights [4][Cout][4*Cin])
void conv3x3(Run& r, const void* x, int B, int Hin, int Win, int Cin, const void* W, int Cout, void* y, int Hout, int Wout, int stride,
             int pad, int upsample, const GemmOpt& o = GemmOpt()) {
    asd_gemm_args g = asd_gemm_args{};
    const int taps = upsample == 3 ? 4 : 9;
    g.A = x; g.W = W; g.C = y; g.M = B * Hout * Wout; g.N = Cout; g.K = taps * Cin; g.lda = 0; g.ldw = taps * Cin; g.ldc = Cout;
    g.bias = o.bias; g.row_bias = o.row_bias; g.rows_per_group = o.row_bias ? o.rows_per_group : 1; g.ld_row_bias = o.ld_row_bias;
    g.residual = o.residual; g.ldr = o.ldr; g.act = o.act; g.out_f32 = o.out_f32;
    g.conv = 1; g.Hin = Hin; g.Win = Win; g.Cin = Cin; g.Hout = Hout; g.Wout = Wout; g.stride = stride; g.pad = pad; g.upsample = upsample;
    g.gn_rows = o.gn_rows;
    if (o.gn_x) set_gn_bwd(g, o);
    if (o.out) { *o.out = Act{}; o.out->p = (const half_t*)y; }
    launch_gemm(r, g, o.out, o.gn_bwd_form, o.gn_next);
}

#define LEAF(call)                                              \
    do {                                                        \
        if (!r.dry && r.status == ASD_OK) {                     \
            const int st__ = (call);                            \
            if (st__ != ASD_OK) r.status = st__;                \
        }                                                       \
    } while (0)

// GroupNorm(32)(+SiLU) of x1 (|| x2 along channels); returns y, optionally the statistics buffer (kept for a backward pass)
half_t* groupnorm(Run& r, const void* x1, int c1, const void* x2, int c2, int B, int hw, const half_t* gamma, const half_t* beta, float eps,
                  int silu, half_t* y = nullptr, float** stats_out = nullptr, const Act* src = nullptr) {
    if (src && src->applied && !x2 && !y && !stats_out && src->applied_gamma == gamma && src->applied_beta == beta && src->applied_eps == eps &&
        src->applied_silu == silu)
        return const_cast<half_t*>(src->applied);          // the producer's split-K reduction already wrote GroupNorm(x)
    if (!y) y = r.mem.halfs((size_t)B * hw * (c1 + c2));
    float* stats = r.mem.floats(ASD_GN_STATS_FLOATS(B));
    if (src && src->nrec > 0 && !x2) {       // the producer's epilogue already reduced the statistics: no pass over x for them
        LEAF(asd_groupnorm_apply_f16(x1, c1, B, hw, gamma, beta, eps, silu, src->rec, src->nrec, y, stats, r.stream));
    } else
    LEAF(asd_groupnorm_f16(x1, c1, x2, c2, B, hw, gamma, beta, eps, silu, y, stats, r.stream));
    if (stats_out) *stats_out = stats;
    return y;
}

// ================================================================================================================================
// UNet
// ================================================================================================================================
struct ULayer { int kind; std::string name; int cin, cout; };   // kind: 0 conv, 1 res, 2 attn, 3 down, 4 up
struct UBlock { std::vector<ULayer> layers; };

struct UNet : Net {
    asd_unet_desc d;
    std::vector<UBlock> inputs, outputs;
    UBlock middle;
    std::map<std::string, int> emb_off, ctx_off;     // column offsets into emb_all / ctx_{k,v}_all
    int emb_total = 0, ctx_total = 0;
};

#define ASD_LN_FOLD_MIN_C 1024      // transformer blocks at least this wide fold their LayerNorms (weights.py: LN_FOLD_MIN_C)
void u_norm(UNet& n, const std::string& p, int c) { n.add(p + ".weight", 1, c); n.add(p + ".bias", 1, c); }
void u_lin(UNet& n, const std::string& p, int cin, int cout, bool bias = true) { n.add(p + ".weight", cout, cin); if (bias) n.add(p + ".bias", 1, cout); }
void u_conv(UNet& n, const std::string& p, int cin, int cout) { n.add(p + ".weight", cout, 9 * pad32(cin)); n.add(p + ".bias", 1, cout); }

void u_res(UNet& n, const std::string& p, int cin, int cout) {
    u_norm(n, p + ".in_layers.0", cin);
    u_conv(n, p + ".in_layers.2", cin, cout);
    u_norm(n, p + ".out_layers.0", cout);
    u_conv(n, p + ".out_layers.3", cout, cout);
    if (cin != cout) u_lin(n, p + ".skip_connection", cin, cout);
    n.emb_off[p] = n.emb_total;           // its emb_layers.1 lives in emb_all (all ResBlocks' projections are one GEMM per forward)
    n.emb_total += cout;
}
void u_attn(UNet& n, const std::string& p, int c) {
    u_norm(n, p + ".norm", c);
    u_lin(n, p + ".proj_in", c, c);
    for (int dd = 0; dd < n.d.transformer_depth; ++dd) {
        const std::string b = p + ".transformer_blocks." + std::to_string(dd);
        // Blocks of the low-resolution levels (c >= ASD_LN_FOLD_MIN_C: 16x16 and 8x8 tokens, M <= 1280 rows at batch 5) fold norm1 / norm2 /
        // norm3 into the GEMMs that consume them (asd_gemm_args.ln_mode): their weights are gamma (.) W and every such GEMM has a "ln_sc"
        // entry = fp32 {rowsum(gamma (.) W), W beta} (stored as 4 halfs per output row; weights.pack_unet applies the same rule).  The fold
        // re-derives the row statistics in every N tile of a row block; measured on the step: it removes a ~5 us LayerNorm launch per
        // norm where a row block has few N tiles and the launch is latency bound, and LOSES where M = 20480 / 5120 (the 2560-column GEGLU
        // projection has 40 N tiles per row block: 64.6 -> 130 us), so the wide levels keep the LayerNorm kernel.
        const bool fold = c >= ASD_LN_FOLD_MIN_C;
        if (!fold) for (const char* nm : {".norm1", ".norm2", ".norm3"}) u_norm(n, b + nm, c);
        n.add(b + ".attn1.to_qk.weight", 2 * c, c);            // q | k rows fused
        if (fold) n.add(b + ".attn1.to_qk.ln_sc", 1, 4 * 2 * c);
        n.add(b + ".attn1.to_v.weight", c, c);
        if (fold) n.add(b + ".attn1.to_v.ln_sc", 1, 4 * c);
        u_lin(n, b + ".attn1.to_out.0", c, c);
        n.add(b + ".attn2.to_q.weight", c, c);
        if (fold) n.add(b + ".attn2.to_q.ln_sc", 1, 4 * c);
        u_lin(n, b + ".attn2.to_out.0", c, c);
        u_lin(n, b + ".ff.net.0.proj", c, 8 * c);              // GEGLU rows interleaved [16 value | 16 gate] (asd_gemm_args.act = 2)
        if (fold) n.add(b + ".ff.net.0.proj.ln_sc", 1, 4 * 8 * c);
        u_lin(n, b + ".ff.net.2", 4 * c, c);
        n.ctx_off[b + ".attn2"] = n.ctx_total;                 // its K / V projections of the context live in ctx_{k,v}_all
        n.ctx_total += c;
    }
    u_lin(n, p + ".proj_out", c, c);
}

// the construction loop of UNetModel.__init__ (openaimodel.py:563-752)
void unet_build(UNet& n) {
    const asd_unet_desc& d = n.d;
    const int mc = d.model_channels, emb = 4 * mc;
    u_lin(n, "time_embed.0", mc, emb);
    u_lin(n, "time_embed.2", emb, emb);
    if (d.camera_dim > 0) { u_lin(n, "camera_embed.0", d.camera_dim, emb); u_lin(n, "camera_embed.2", emb, emb); }
    u_conv(n, "input_blocks.0.0", d.in_channels, mc);
    n.inputs.push_back(UBlock{{ULayer{0, "input_blocks.0.0", d.in_channels, mc}}});
    std::vector<int> chans{mc};
    int ch = mc, ds = 0;                                        // ds = log2 of the downsample factor
    for (int lvl = 0; lvl < d.n_levels; ++lvl) {
        const int mult = d.channel_mult[lvl];
        for (int k = 0; k < d.num_res_blocks; ++k) {
            const std::string p = "input_blocks." + std::to_string(n.inputs.size());
            UBlock b;
            u_res(n, p + ".0", ch, mult * mc);
            b.layers.push_back(ULayer{1, p + ".0", ch, mult * mc});
            ch = mult * mc;
            if (d.attention_ds_mask & (1 << ds)) { u_attn(n, p + ".1", ch); b.layers.push_back(ULayer{2, p + ".1", ch, ch}); }
            n.inputs.push_back(b);
            chans.push_back(ch);
        }
        if (lvl != d.n_levels - 1) {
            const std::string p = "input_blocks." + std::to_string(n.inputs.size()) + ".0.op";
            u_conv(n, p, ch, ch);
            n.inputs.push_back(UBlock{{ULayer{3, p, ch, ch}}});
            chans.push_back(ch);
            ++ds;
        }
    }
    u_res(n, "middle_block.0", ch, ch);
    u_attn(n, "middle_block.1", ch);
    u_res(n, "middle_block.2", ch, ch);
    n.middle.layers = {ULayer{1, "middle_block.0", ch, ch}, ULayer{2, "middle_block.1", ch, ch}, ULayer{1, "middle_block.2", ch, ch}};
    for (int lvl = d.n_levels - 1; lvl >= 0; --lvl) {
        const int mult = d.channel_mult[lvl];
        for (int i = 0; i <= d.num_res_blocks; ++i) {
            const int ich = chans.back();
            chans.pop_back();
            const std::string p = "output_blocks." + std::to_string(n.outputs.size());
            UBlock b;
            u_res(n, p + ".0", ch + ich, mc * mult);
            b.layers.push_back(ULayer{1, p + ".0", ch + ich, mc * mult});
            ch = mc * mult;
            int k = 1;
            if (d.attention_ds_mask & (1 << ds)) {
                u_attn(n, p + "." + std::to_string(k), ch);
                b.layers.push_back(ULayer{2, p + "." + std::to_string(k), ch, ch});
                ++k;
            }
            if (lvl && i == d.num_res_blocks) {
                const std::string q = p + "." + std::to_string(k) + ".conv";
                // the upsampling convolution in its parity form: [4 parities][cout][2 x 2 taps x cin] pre-summed taps (weights.pack_unet)
                n.add(q + ".weight", 4 * ch, 4 * pad32(ch)); n.add(q + ".bias", 1, ch);
                b.layers.push_back(ULayer{4, q, ch, ch});
                --ds;
            }
            n.outputs.push_back(b);
        }
    }
    u_norm(n, "out.0", ch);
    u_conv(n, "out.2", mc, d.out_channels);
    n.add("emb_all.weight", n.emb_total, emb);
    n.add("emb_all.bias", 1, n.emb_total);
    n.add("ctx_k_all.weight", n.ctx_total, d.context_dim);
    n.add("ctx_v_all.weight", n.ctx_total, d.context_dim);
}

struct UState {             // per-forward quantities shared by the layers
    int B, F, n_ctx, ctx_stride;
    const half_t* emb_all; int emb_ld;
    const half_t* k_all; const half_t* vT_all;
    // shared-input prefix (asd_unet_fwd_shared): while `full` is set this state describes the batch of DISTINCT inputs; the first
    // transformer switches to *full after its first self-attention (the first layer that reads the text context is the cross-
    // attention behind it) by broadcasting its activations with `expand` (int32 [full->B] on the device)
    const UState* full = nullptr;
    const int* expand = nullptr;
};

// rows of `row_halfs` fp16 values: dst[i] = src[idx[i]]
half_t* gather_rows(Run& r, const half_t* src, const int* idx, int n_out, size_t row_halfs) {
    half_t* dst = r.mem.halfs((size_t)n_out * row_halfs);
    LEAF(asd_gather_rows_f16(src, idx, n_out, (int64_t)row_halfs, dst, r.stream));
    return dst;
}

Act u_resblock(UNet& n, Run& r, const UState& s, const std::string& p, const Act& xin, int cin, int cout, int Hh, int Ww, const GnNext* next = nullptr) {
    const int B = s.B, hw = Hh * Ww, M = B * hw;
    const half_t* x = xin.p;
    half_t* t1 = groupnorm(r, x, cin, nullptr, 0, B, hw, n.w(p + ".in_layers.0.weight"), n.w(p + ".in_layers.0.bias"), 1e-5f, 1, nullptr, nullptr, &xin);
    half_t* t2 = r.mem.halfs((size_t)M * cout);
    Act a2;
    GemmOpt o1;
    o1.bias = n.w(p + ".in_layers.2.bias");
    o1.row_bias = s.emb_all + n.emb_off[p]; o1.rows_per_group = hw; o1.ld_row_bias = s.emb_ld;   // + emb_layers(emb)[:, :, None, None]
    o1.gn_rows = hw; o1.out = &a2;
    const GnNext gn2{n.w(p + ".out_layers.0.weight"), n.w(p + ".out_layers.0.bias"), 1e-5f, 1};
    o1.gn_next = &gn2;
    conv3x3(r, t1, B, Hh, Ww, cin, n.w(p + ".in_layers.2.weight"), cout, t2, Hh, Ww, 1, 1, 0, o1);
    half_t* t3 = groupnorm(r, t2, cout, nullptr, 0, B, hw, n.w(p + ".out_layers.0.weight"), n.w(p + ".out_layers.0.bias"), 1e-5f, 1, nullptr, nullptr, &a2);
    const half_t* skip = x;
    if (n.has(p + ".skip_connection.weight")) {
        half_t* sk = r.mem.halfs((size_t)M * cout);
        GemmOpt os;
        os.bias = n.w(p + ".skip_connection.bias");
        gemm(r, x, M, cin, n.w(p + ".skip_connection.weight"), cout, cin, cin, sk, cout, os);
        skip = sk;
    }
    half_t* out = r.mem.halfs((size_t)M * cout);
    Act ao;
    GemmOpt o2;
    o2.bias = n.w(p + ".out_layers.3.bias"); o2.residual = skip; o2.ldr = cout; o2.gn_rows = hw; o2.out = &ao; o2.gn_next = next;
    conv3x3(r, t3, B, Hh, Ww, cout, n.w(p + ".out_layers.3.weight"), cout, out, Hh, Ww, 1, 1, 0, o2);
    return ao;
}

half_t* layernorm(Run& r, const half_t* x, int rows, int c, const half_t* g, const half_t* b) {
    half_t* y = r.mem.halfs((size_t)rows * c);
    LEAF(asd_layernorm_f16(x, rows, c, g, b, 1e-5f, y, r.stream));
    return y;
}

Act u_transformer(UNet& n, Run& r, const UState& s_in, const std::string& p, const Act& xin, int C, int Hh, int Ww, const GnNext* next = nullptr) {
    const UState* sp = &s_in;
    int B = sp->B, L = Hh * Ww, M = B * L;
    const int heads = C / 64, F = sp->F;
    const half_t* x = xin.p;
    half_t* h = groupnorm(r, x, C, nullptr, 0, B, L, n.w(p + ".norm.weight"), n.w(p + ".norm.bias"), 1e-6f, 0, nullptr, nullptr, &xin);
    {
        half_t* h2 = r.mem.halfs((size_t)M * C);
        GemmOpt o; o.bias = n.w(p + ".proj_in.bias");
        gemm(r, h, M, C, n.w(p + ".proj_in.weight"), C, C, C, h2, C, o);
        h = h2;
    }
    for (int dd = 0; dd < n.d.transformer_depth; ++dd) {
        const std::string b = p + ".transformer_blocks." + std::to_string(dd);
        // self-attention (MVDream: over the F views of a group, attention.py:348-354)
        // Folded blocks: norm1 lives inside the two projections — the q | k GEMM reduces the row statistics of h in its main loop and
        // leaves them for the V^T GEMM, whose W operand is the same h (attention.py:262: attn1(norm1(x))).
        const bool fold = n.has(b + ".attn1.to_qk.ln_sc");
        GemmOpt oqk, ov;
        const half_t* y = h;
        if (fold) {
            float* lnst = r.mem.floats((size_t)M * 2);
            oqk.ln_mode = 1; oqk.ln_sc = (const float*)n.w(b + ".attn1.to_qk.ln_sc"); oqk.ln_stats = lnst;
            ov.ln_mode = 2; ov.ln_sc = (const float*)n.w(b + ".attn1.to_v.ln_sc"); ov.ln_stats = lnst;
        } else y = layernorm(r, h, M, C, n.w(b + ".norm1.weight"), n.w(b + ".norm1.bias"));
        half_t* qk = r.mem.halfs((size_t)M * 2 * C);
        gemm(r, y, M, C, n.w(b + ".attn1.to_qk.weight"), 2 * C, C, C, qk, 2 * C, oqk);
        half_t* vT = r.mem.halfs((size_t)C * M);                              // V^T = W_v norm1(h)^T: operands swapped
        gemm(r, n.w(b + ".attn1.to_v.weight"), C, C, y, M, C, C, vT, M, ov);
        half_t* o1 = r.mem.halfs((size_t)M * C);
        LEAF(asd_attention_f16(qk, 2 * C, qk + C, 2 * C, vT, M, o1, C, B / F, heads, F * L, F * L, F * L, 0.125f, r.zero_page, r.stream));
        half_t* h1 = r.mem.halfs((size_t)M * C);
        { GemmOpt o; o.bias = n.w(b + ".attn1.to_out.0.bias"); o.residual = h; o.ldr = C;
          gemm(r, o1, M, C, n.w(b + ".attn1.to_out.0.weight"), C, C, C, h1, C, o); }
        if (sp->full) {          // end of the shared prefix: every batch entry gets its distinct input's activations
            const UState* f = sp->full;
            h1 = gather_rows(r, h1, sp->expand, f->B, (size_t)L * C);
            x = gather_rows(r, x, sp->expand, f->B, (size_t)L * C);      // the block's residual input (proj_out below)
            sp = f; B = f->B; M = B * L;
        }
        const UState& s = *sp;
        // cross-attention on the text context (K / V^T of every layer were projected once per forward)
        half_t* q = r.mem.halfs((size_t)M * C);
        { GemmOpt o;                                                                                    // attn2(norm2(x))
          y = h1;
          if (fold) { o.ln_mode = 1; o.ln_sc = (const float*)n.w(b + ".attn2.to_q.ln_sc"); }
          else y = layernorm(r, h1, M, C, n.w(b + ".norm2.weight"), n.w(b + ".norm2.bias"));
          gemm(r, y, M, C, n.w(b + ".attn2.to_q.weight"), C, C, C, q, C, o); }
        const int coff = n.ctx_off[b + ".attn2"], ldv = B * s.ctx_stride;
        half_t* o2 = r.mem.halfs((size_t)M * C);
        LEAF(asd_attention_f16(q, C, s.k_all + coff, n.ctx_total, s.vT_all + (size_t)coff * ldv, ldv, o2, C, B, heads, L, s.n_ctx, s.ctx_stride,
                               0.125f, r.zero_page, r.stream));
        half_t* h2 = r.mem.halfs((size_t)M * C);
        { GemmOpt o; o.bias = n.w(b + ".attn2.to_out.0.bias"); o.residual = h1; o.ldr = C;
          gemm(r, o2, M, C, n.w(b + ".attn2.to_out.0.weight"), C, C, C, h2, C, o); }
        // GEGLU feed-forward
        half_t* g = r.mem.halfs((size_t)M * 4 * C);
        { GemmOpt o; o.bias = n.w(b + ".ff.net.0.proj.bias"); o.act = 2;                               // ff(norm3(x))
          y = h2;
          if (fold) { o.ln_mode = 1; o.ln_sc = (const float*)n.w(b + ".ff.net.0.proj.ln_sc"); }
          else y = layernorm(r, h2, M, C, n.w(b + ".norm3.weight"), n.w(b + ".norm3.bias"));
          gemm(r, y, M, C, n.w(b + ".ff.net.0.proj.weight"), 8 * C, C, C, g, 4 * C, o); }
        half_t* h3 = r.mem.halfs((size_t)M * C);
        { GemmOpt o; o.bias = n.w(b + ".ff.net.2.bias"); o.residual = h2; o.ldr = C;
          gemm(r, g, M, 4 * C, n.w(b + ".ff.net.2.weight"), C, 4 * C, 4 * C, h3, C, o); }
        h = h3;
    }
    half_t* out = r.mem.halfs((size_t)M * C);
    Act ao;
    GemmOpt o; o.bias = n.w(p + ".proj_out.bias"); o.residual = x; o.ldr = C; o.gn_rows = L; o.out = &ao; o.gn_next = next;
    gemm(r, h, M, C, n.w(p + ".proj_out.weight"), C, C, C, out, C, o);
    return ao;
}

// the GroupNorm a layer opens with (ResBlock in_layers.0: GroupNorm32 + SiLU; SpatialTransformer.norm: GroupNorm, eps 1e-6), if any
bool u_first_norm(UNet& n, const ULayer* l, GnNext* out) {
    if (!l) return false;
    if (l->kind == 1) { *out = GnNext{n.w(l->name + ".in_layers.0.weight"), n.w(l->name + ".in_layers.0.bias"), 1e-5f, 1}; return true; }
    if (l->kind == 2) { *out = GnNext{n.w(l->name + ".norm.weight"), n.w(l->name + ".norm.bias"), 1e-6f, 0}; return true; }
    return false;
}

// `after`: the layer that consumes this block's output directly (no concatenation in between), when the caller knows it
Act u_apply(UNet& n, Run& r, const UState& s, const UBlock& blk, Act h, int* Hh, int* Ww, const ULayer* after = nullptr) {
    for (size_t li = 0; li < blk.layers.size(); ++li) {
        const ULayer& l = blk.layers[li];
        const int B = s.B;
        GnNext gnn;
        const GnNext* next = u_first_norm(n, li + 1 < blk.layers.size() ? &blk.layers[li + 1] : after, &gnn) ? &gnn : nullptr;
        if (l.kind == 0) {
            half_t* y = r.mem.halfs((size_t)B * *Hh * *Ww * l.cout);
            Act a;
            GemmOpt o; o.bias = n.w(l.name + ".bias"); o.gn_rows = *Hh * *Ww; o.out = &a;
            conv3x3(r, h.p, B, *Hh, *Ww, pad32(l.cin), n.w(l.name + ".weight"), l.cout, y, *Hh, *Ww, 1, 1, 0, o);
            h = a;
        } else if (l.kind == 1) {
            h = u_resblock(n, r, s, l.name, h, l.cin, l.cout, *Hh, *Ww, next);
        } else if (l.kind == 2) {
            h = u_transformer(n, r, s, l.name, h, l.cout, *Hh, *Ww, next);
        } else if (l.kind == 3) {
            const int Ho = (*Hh + 2 - 3) / 2 + 1, Wo = (*Ww + 2 - 3) / 2 + 1;
            half_t* y = r.mem.halfs((size_t)B * Ho * Wo * l.cout);
            Act a;
            GemmOpt o; o.bias = n.w(l.name + ".bias"); o.gn_rows = Ho * Wo; o.out = &a; o.gn_next = next;
            conv3x3(r, h.p, B, *Hh, *Ww, l.cin, n.w(l.name + ".weight"), l.cout, y, Ho, Wo, 2, 1, 0, o);
            h = a; *Hh = Ho; *Ww = Wo;
        } else {
            half_t* y = r.mem.halfs((size_t)B * 4 * *Hh * *Ww * l.cout);
            Act a;
            GemmOpt o; o.bias = n.w(l.name + ".bias"); o.gn_rows = 4 * *Hh * *Ww; o.out = &a;
            conv3x3(r, h.p, B, *Hh, *Ww, l.cin, n.w(l.name + ".weight"), l.cout, y, 2 * *Hh, 2 * *Ww, 1, 1, 3, o);
            h = a; *Hh *= 2; *Ww *= 2;
        }
    }
    return h;
}

// UNetModel.forward (openaimodel.py:771-808) / MultiViewUNetModel.forward (:1175-1213)
void unet_run(UNet& n, Run& r, const half_t* x, const float* t, const half_t* ctx, const half_t* camera, int B, int Hh, int Ww, int n_ctx,
              int num_frames, float* eps, const int* uniq_src = nullptr, const int* expand = nullptr, int Bu = 0) {
    const asd_unet_desc& d = n.d;
    const int mc = d.model_channels, emb = 4 * mc;
    UState s;
    s.B = B; s.F = d.camera_dim > 0 ? num_frames : 1; s.n_ctx = n_ctx; s.ctx_stride = (n_ctx + 7) / 8 * 8;
    half_t* t_emb = r.mem.halfs((size_t)B * mc);
    LEAF(asd_timestep_embedding_f16(t, B, mc, t_emb, r.stream));
    half_t* e0 = r.mem.halfs((size_t)B * emb);
    { GemmOpt o; o.bias = n.w("time_embed.0.bias"); o.act = 1; gemm(r, t_emb, B, mc, n.w("time_embed.0.weight"), emb, mc, mc, e0, emb, o); }
    half_t* e = r.mem.halfs((size_t)B * emb);
    { GemmOpt o; o.bias = n.w("time_embed.2.bias"); gemm(r, e0, B, emb, n.w("time_embed.2.weight"), emb, emb, emb, e, emb, o); }
    if (d.camera_dim > 0) {   // emb += camera_embed(camera)
        half_t* c0 = r.mem.halfs((size_t)B * emb);
        { GemmOpt o; o.bias = n.w("camera_embed.0.bias"); o.act = 1;
          gemm(r, camera, B, d.camera_dim, n.w("camera_embed.0.weight"), emb, d.camera_dim, d.camera_dim, c0, emb, o); }
        half_t* e2 = r.mem.halfs((size_t)B * emb);
        { GemmOpt o; o.bias = n.w("camera_embed.2.bias"); o.residual = e; o.ldr = emb;
          gemm(r, c0, B, emb, n.w("camera_embed.2.weight"), emb, emb, emb, e2, emb, o); }
        e = e2;
    }
    half_t* es = r.mem.halfs((size_t)B * emb);
    LEAF(asd_silu_f16(e, (int64_t)B * emb, es, r.stream));
    half_t* emb_all = r.mem.halfs((size_t)B * n.emb_total);     // every ResBlock's emb_layers at once
    { GemmOpt o; o.bias = n.w("emb_all.bias"); gemm(r, es, B, emb, n.w("emb_all.weight"), n.emb_total, emb, emb, emb_all, n.emb_total, o); }
    s.emb_all = emb_all; s.emb_ld = n.emb_total;
    const int rows_ctx = B * s.ctx_stride;
    half_t* k_all = r.mem.halfs((size_t)rows_ctx * n.ctx_total);
    gemm(r, ctx, rows_ctx, d.context_dim, n.w("ctx_k_all.weight"), n.ctx_total, d.context_dim, d.context_dim, k_all, n.ctx_total);
    half_t* vT_all = r.mem.halfs((size_t)n.ctx_total * rows_ctx);
    gemm(r, n.w("ctx_v_all.weight"), n.ctx_total, d.context_dim, ctx, rows_ctx, d.context_dim, d.context_dim, vT_all, rows_ctx);
    s.k_all = k_all; s.vT_all = vT_all;

    struct Skip { const half_t* p; int c; };
    std::vector<Skip> hs;
    Act h;
    h.p = x;
    int hh = Hh, ww = Ww;
    size_t first = 0;
    if (Bu > 0 && Bu < B && !n.inputs.empty()) {
        // shared-input prefix: conv_in (and, when the second block opens with ResBlock + transformer, everything up to its first
        // cross-attention) on the Bu distinct inputs only
        UState su = s;
        su.B = Bu; su.full = &s; su.expand = expand;
        su.emb_all = gather_rows(r, emb_all, uniq_src, Bu, (size_t)n.emb_total);
        Act hu;
        hu.p = gather_rows(r, x, uniq_src, Bu, (size_t)Hh * Ww * 32);
        hu = u_apply(n, r, su, n.inputs[0], hu, &hh, &ww);
        const int c0 = n.inputs[0].layers.back().cout;
        bool has_attn = false;
        if (n.inputs.size() > 1)
            for (const ULayer& l : n.inputs[1].layers) has_attn = has_attn || l.kind == 2;
        h = Act{gather_rows(r, hu.p, expand, B, (size_t)hh * ww * c0), nullptr, 0};
        hs.push_back(Skip{h.p, c0});
        first = 1;
        if (has_attn && n.inputs[1].layers.front().kind != 3 && n.inputs[1].layers.front().kind != 4) {
            h = u_apply(n, r, su, n.inputs[1], hu, &hh, &ww);        // leaves the prefix inside its first transformer: full batch
            hs.push_back(Skip{h.p, n.inputs[1].layers.back().cout});
            first = 2;
        }
    }
    for (size_t bi = first; bi < n.inputs.size(); ++bi) {
        const UBlock& b = n.inputs[bi];
        const UBlock& nb = bi + 1 < n.inputs.size() ? n.inputs[bi + 1] : n.middle;
        h = u_apply(n, r, s, b, h, &hh, &ww, nb.layers.empty() ? nullptr : &nb.layers.front());
        hs.push_back(Skip{h.p, b.layers.back().cout});
    }
    h = u_apply(n, r, s, n.middle, h, &hh, &ww);
    int ch = n.middle.layers.back().cout;
    for (const UBlock& b : n.outputs) {
        const Skip sk = hs.back();
        hs.pop_back();
        half_t* cat = r.mem.halfs((size_t)B * hh * ww * (ch + sk.c));
        LEAF(asd_concat_f16(h.p, ch, sk.p, sk.c, (int64_t)B * hh * ww, cat, r.stream));
        Act hc;          // a concatenation: the GroupNorm groups span both halves, so the halves' records do not apply
        hc.p = cat;
        h = u_apply(n, r, s, b, hc, &hh, &ww);
        ch = b.layers.back().cout;
    }
    half_t* y = groupnorm(r, h.p, ch, nullptr, 0, B, hh * ww, n.w("out.0.weight"), n.w("out.0.bias"), 1e-5f, 1, nullptr, nullptr, &h);
    GemmOpt o; o.bias = n.w("out.2.bias"); o.out_f32 = 1;
    conv3x3(r, y, B, hh, ww, ch, n.w("out.2.weight"), d.out_channels, eps, hh, ww, 1, 1, 0, o);
}

// ================================================================================================================================
// VAE encoder: forward with the activations its input gradient needs kept in the workspace, and that backward pass
// ================================================================================================================================
struct VLayer { int kind; std::string name; int cin, cout; };   // 0 conv_in, 1 res, 2 down, 3 attn, 4 out (norm_out + conv_out*quant_conv)

struct Vae : Net {
    asd_vae_desc d;
    std::vector<VLayer> plan;
    std::map<const void*, size_t> scratch_of;      // workspace -> split-K scratch size its last forward laid it out with
    std::map<const void*, std::vector<int>> recs_of;   // workspace -> GroupNorm record counts of that forward, in allocation order
    // the last shape whose forward + backward layout was sized and found to fit (valid while the plan table does not change)
    struct Sized { int batch = 0, H = 0, W = 0, tune = 0; uint64_t gen = 0; int64_t bytes = 0, need = 0; size_t scratch = 0; } sized;
    std::map<const void*, uint64_t> gen_of;        // workspace -> plan generation of its last forward
};

// conv: forward weights [cout, 9*pad32(cin)], input-gradient weights [pad32(cin), 9*pad32(cout)] (roles swapped, taps flipped for
// stride 1; the stride-2 gradient is FOUR 2x2 convolutions over the low-resolution gradient image, one per input-pixel parity — the
// kernel's upsample == 3 form, weights.py: _pack_stride2_dgrad_conv3x3), bias
void v_conv(Vae& n, const std::string& p, int cin, int cout, bool stride2 = false) {
    n.add(p + ".fwd", cout, 9 * pad32(cin));
    if (stride2) n.add(p + ".bwd", 4 * pad32(cin), 4 * pad32(cout));    // parity form: [4 parities][cin][2 x 2 tap slots x cout]
    else n.add(p + ".bwd", pad32(cin), 9 * pad32(cout));
    n.add(p + ".bias", 1, cout);
}
void v_norm(Vae& n, const std::string& p, int c) { n.add(p + ".weight", 1, c); n.add(p + ".bias", 1, c); }
void v_res(Vae& n, const std::string& p, int cin, int cout) {
    v_norm(n, p + ".norm1", cin);
    v_conv(n, p + ".conv1", cin, cout);
    v_norm(n, p + ".norm2", cout);
    v_conv(n, p + ".conv2", cout, cout);
    if (cin != cout) { n.add(p + ".nin.w", cout, cin); n.add(p + ".nin.wt", cin, cout); n.add(p + ".nin.b", 1, cout); }
    n.plan.push_back(VLayer{1, p, cin, cout});
}

void vae_build(Vae& n) {
    const asd_vae_desc& d = n.d;
    v_conv(n, "encoder.conv_in", d.in_channels, d.ch);
    n.plan.push_back(VLayer{0, "encoder.conv_in", d.in_channels, d.ch});
    int block_in = d.ch;
    for (int lvl = 0; lvl < d.n_levels; ++lvl) {
        const int block_out = d.ch * d.ch_mult[lvl];
        for (int b = 0; b < d.num_res_blocks; ++b) {
            v_res(n, "encoder.down." + std::to_string(lvl) + ".block." + std::to_string(b), block_in, block_out);
            block_in = block_out;
        }
        if (lvl != d.n_levels - 1) {
            const std::string p = "encoder.down." + std::to_string(lvl) + ".downsample.conv";
            v_conv(n, p, block_in, block_in, true);
            n.plan.push_back(VLayer{2, p, block_in, block_in});
        }
    }
    v_res(n, "encoder.mid.block_1", block_in, block_in);
    {
        const std::string p = "encoder.mid.attn_1";
        v_norm(n, p + ".norm", block_in);
        for (const char* nm : {".q", ".k", ".v", ".proj_out"}) {
            n.add(p + nm + ".w", block_in, block_in); n.add(p + nm + ".wt", block_in, block_in); n.add(p + nm + ".b", 1, block_in);
        }
        n.plan.push_back(VLayer{3, p, block_in, block_in});
    }
    v_res(n, "encoder.mid.block_2", block_in, block_in);
    v_norm(n, "encoder.norm_out", block_in);
    v_conv(n, "encoder.conv_out_quant", block_in, 2 * d.embed_dim);   // quant_conv o conv_out composed at pack time (both linear)
    n.plan.push_back(VLayer{4, "encoder", block_in, 2 * d.embed_dim});
}

// Buffers a forward leaves behind for the input-gradient pass.  The backward pass re-derives the very same addresses by
// replaying the allocation sequence (vae_forward with `launch == false`), so nothing but the workspace travels between the calls.
struct VSaved {
    const half_t* x = nullptr;        // layer input
    half_t* t2 = nullptr;             // ResnetBlock: conv1 output (input of norm2)
    float* st1 = nullptr; float* st2 = nullptr;
    half_t *q = nullptr, *k = nullptr, *v = nullptr, *p = nullptr;   // attention: projections and softmax probabilities [B][L,L]
    half_t* out = nullptr;
    int H = 0, W = 0;
};

void vae_forward(Vae& n, Run& r, const half_t* x32, int B, int Hh, int Ww, float* moments, std::vector<VSaved>& saved, bool launch) {
    const bool was_dry = r.dry;
    if (!launch) r.dry = true;
    // two scratch activations for GroupNorm outputs (consumed at once by the following convolution)
    size_t max_act = 0;
    {
        int hh = Hh, ww = Ww;
        for (const VLayer& l : n.plan) {
            const size_t a = (size_t)B * hh * ww * (l.cin > l.cout ? l.cin : l.cout);
            if (a > max_act) max_act = a;
            if (l.kind == 2) { hh = (hh + 1 - 3) / 2 + 1; ww = (ww + 1 - 3) / 2 + 1; }
        }
    }
    half_t* tmp = r.mem.halfs(max_act);
    saved.assign(n.plan.size(), VSaved());
    Act h;
    h.p = x32;
    int hh = Hh, ww = Ww;
    for (size_t i = 0; i < n.plan.size(); ++i) {
        const VLayer& l = n.plan[i];
        VSaved& sv = saved[i];
        Act produced;                               // this layer's output and, when its producer could, its GroupNorm records
        sv.x = h.p; sv.H = hh; sv.W = ww;
        const int hw = hh * ww, M = B * hw;
        if (l.kind == 0) {
            sv.out = r.mem.halfs((size_t)M * l.cout);
            GemmOpt o; o.bias = n.w(l.name + ".bias"); o.gn_rows = hw; o.out = &produced;
            conv3x3(r, h.p, B, hh, ww, pad32(l.cin), n.w(l.name + ".fwd"), l.cout, sv.out, hh, ww, 1, 1, 0, o);
        } else if (l.kind == 1) {
            const std::string& p = l.name;
            groupnorm(r, h.p, l.cin, nullptr, 0, B, hw, n.w(p + ".norm1.weight"), n.w(p + ".norm1.bias"), 1e-6f, 1, tmp, &sv.st1, &h);
            sv.t2 = r.mem.halfs((size_t)M * l.cout);
            Act a2;
            { GemmOpt o; o.bias = n.w(p + ".conv1.bias"); o.gn_rows = hw; o.out = &a2;
              conv3x3(r, tmp, B, hh, ww, l.cin, n.w(p + ".conv1.fwd"), l.cout, sv.t2, hh, ww, 1, 1, 0, o); }
            groupnorm(r, sv.t2, l.cout, nullptr, 0, B, hw, n.w(p + ".norm2.weight"), n.w(p + ".norm2.bias"), 1e-6f, 1, tmp, &sv.st2, &a2);
            const half_t* sc = h.p;
            if (n.has(p + ".nin.w")) {
                half_t* s2 = r.mem.halfs((size_t)M * l.cout);
                GemmOpt o; o.bias = n.w(p + ".nin.b");
                gemm(r, h.p, M, l.cin, n.w(p + ".nin.w"), l.cout, l.cin, l.cin, s2, l.cout, o);
                sc = s2;
            }
            sv.out = r.mem.halfs((size_t)M * l.cout);
            GemmOpt o; o.bias = n.w(p + ".conv2.bias"); o.residual = sc; o.ldr = l.cout;   // x + h in the conv epilogue (model.py:141-148)
            o.gn_rows = hw; o.out = &produced;
            conv3x3(r, tmp, B, hh, ww, l.cout, n.w(p + ".conv2.fwd"), l.cout, sv.out, hh, ww, 1, 1, 0, o);
        } else if (l.kind == 2) {          // stride 2 with asymmetric (0,1,0,1) zero padding (model.py:80-85)
            const int ho = (hh + 1 - 3) / 2 + 1, wo = (ww + 1 - 3) / 2 + 1;
            sv.out = r.mem.halfs((size_t)B * ho * wo * l.cout);
            GemmOpt o; o.bias = n.w(l.name + ".bias"); o.gn_rows = ho * wo; o.out = &produced;
            conv3x3(r, h.p, B, hh, ww, l.cin, n.w(l.name + ".fwd"), l.cout, sv.out, ho, wo, 2, 0, 0, o);
            hh = ho; ww = wo;
        } else if (l.kind == 3) {          // single-head attention over the hw positions of each image
            const std::string& p = l.name;
            const int C = l.cin, L = hw;
            groupnorm(r, h.p, C, nullptr, 0, B, hw, n.w(p + ".norm.weight"), n.w(p + ".norm.bias"), 1e-6f, 0, tmp, &sv.st1, &h);
            sv.q = r.mem.halfs((size_t)M * C); sv.k = r.mem.halfs((size_t)M * C); sv.v = r.mem.halfs((size_t)M * C);
            half_t* dst[3] = {sv.q, sv.k, sv.v};
            const char* nm[3] = {".q", ".k", ".v"};
            for (int j = 0; j < 3; ++j) {
                GemmOpt o; o.bias = n.w(p + nm[j] + ".b");
                gemm(r, tmp, M, C, n.w(p + nm[j] + ".w"), C, C, C, dst[j], C, o);
            }
            sv.p = r.mem.halfs((size_t)B * L * L);
            half_t* sc = r.mem.halfs((size_t)L * L);
            half_t* vt = r.mem.halfs((size_t)C * L);
            half_t* ao = r.mem.halfs((size_t)M * C);
            for (int b = 0; b < B; ++b) {
                const size_t ro = (size_t)b * L * C;
                half_t* P = sv.p + (size_t)b * L * L;
                gemm(r, sv.q + ro, L, C, sv.k + ro, L, C, C, sc, L);                                   // S = Q K^T
                LEAF(asd_softmax_f16(sc, L, L, L, 1.0f / sqrtf((float)C), P, L, r.stream));
                LEAF(asd_transpose_f16(sv.v + ro, L, C, C, vt, L, r.stream));
                gemm(r, P, L, L, vt, C, L, L, ao + ro, C);                                             // O = P V
            }
            sv.out = r.mem.halfs((size_t)M * C);
            GemmOpt o; o.bias = n.w(p + ".proj_out.b"); o.residual = h.p; o.ldr = C;                    // x + proj_out(attn)
            o.gn_rows = hw; o.out = &produced;
            gemm(r, ao, M, C, n.w(p + ".proj_out.w"), C, C, C, sv.out, C, o);
        } else {
            groupnorm(r, h.p, l.cin, nullptr, 0, B, hw, n.w(l.name + ".norm_out.weight"), n.w(l.name + ".norm_out.bias"), 1e-6f, 1, tmp, &sv.st1, &h);
            GemmOpt o; o.bias = n.w(l.name + ".conv_out_quant.bias"); o.out_f32 = 1;
            conv3x3(r, tmp, B, hh, ww, l.cin, n.w(l.name + ".conv_out_quant.fwd"), l.cout, moments, hh, ww, 1, 1, 0, o);
            sv.out = nullptr;
        }
        if (sv.out) { h = produced; h.p = sv.out; }
    }
    r.dry = was_dry;
}

half_t* gn_bwd(Run& r, const half_t* x, const half_t* dy, int c, int B, int hw, const half_t* g, const half_t* b, int silu, const float* st,
               const half_t* dx_add, half_t* dx, const Act* src = nullptr) {
    float* bst = r.mem.floats(ASD_GN_STATS_FLOATS(B) - 64 * B);
    if (src && src->nrec > 0) {          // the conv / GEMM that produced dy left the two reductions behind: no pass over (x, dy) for them
        LEAF(asd_groupnorm_bwd_apply_f16(x, dy, c, B, hw, g, b, 1e-6f, silu, st, src->rec, src->nrec, dx_add, dx, bst, r.stream));
    } else
    LEAF(asd_groupnorm_bwd_f16(x, dy, c, B, hw, g, b, 1e-6f, silu, st, dx_add, dx, bst, r.stream));
    return dx;
}

// options of a launch whose output is the gradient reaching GroupNorm(x) [+ SiLU]
GemmOpt feeds_gn_bwd(Act* out, int hw, const half_t* x, const float* fstats, const half_t* gamma, const half_t* beta, int silu) {
    GemmOpt o;
    o.gn_rows = hw; o.out = out; o.gn_x = x; o.gn_fstats = fstats; o.gn_gamma = gamma; o.gn_beta = beta; o.gn_silu = silu;
    o.gn_bwd_form = true;
    return o;
}

void vae_backward(Vae& n, Run& r, const std::vector<VSaved>& saved, const float* d_moments, int B, half_t* dx32) {
    // gradient buffers: four rotating activations of the largest size
    size_t max_act = 0;
    for (size_t i = 0; i < n.plan.size(); ++i) {
        const VLayer& l = n.plan[i];
        const size_t a = (size_t)B * saved[i].H * saved[i].W * (l.cin > l.cout ? l.cin : l.cout);
        if (a > max_act) max_act = a;
    }
    half_t* buf[4];
    for (int j = 0; j < 4; ++j) buf[j] = r.mem.halfs(max_act);
    int cur = 0;                                         // buf[cur] holds the gradient w.r.t. the current layer's OUTPUT
    auto other = [&](int k) { return buf[(cur + k) & 3]; };
    const half_t* dy = nullptr;
    for (int i = (int)n.plan.size() - 1; i >= 0; --i) {
        const VLayer& l = n.plan[i];
        const VSaved& sv = saved[i];
        const int hh = sv.H, ww = sv.W, hw = hh * ww, M = B * hw;
        if (l.kind == 4) {
            const int cp = pad32(l.cout);
            half_t* dm = other(1);
            LEAF(asd_pad_cast_f16(d_moments, M, l.cout, dm, cp, r.stream));
            half_t* dt = other(2);
            Act at;
            conv3x3(r, dm, B, hh, ww, cp, n.w(l.name + ".conv_out_quant.bwd"), l.cin, dt, hh, ww, 1, 1, 0,
                    feeds_gn_bwd(&at, hw, sv.x, sv.st1, n.w(l.name + ".norm_out.weight"), n.w(l.name + ".norm_out.bias"), 1));
            gn_bwd(r, sv.x, dt, l.cin, B, hw, n.w(l.name + ".norm_out.weight"), n.w(l.name + ".norm_out.bias"), 1, sv.st1, nullptr, buf[cur], &at);
            dy = buf[cur];
        } else if (l.kind == 3) {
            const std::string& p = l.name;
            const int C = l.cin, L = hw;
            half_t* d_ao = other(1);
            gemm(r, dy, M, C, n.w(p + ".proj_out.wt"), C, C, C, d_ao, C);
            half_t* dq = r.mem.halfs((size_t)M * C); half_t* dk = r.mem.halfs((size_t)M * C); half_t* dv = r.mem.halfs((size_t)M * C);
            half_t* pt = r.mem.halfs((size_t)L * L); half_t* dp = r.mem.halfs((size_t)L * L); half_t* dsm = r.mem.halfs((size_t)L * L);
            half_t* tA = r.mem.halfs((size_t)C * L);
            for (int b = 0; b < B; ++b) {
                const size_t ro = (size_t)b * L * C;
                const half_t* P = sv.p + (size_t)b * L * L;
                LEAF(asd_transpose_f16(P, L, L, L, pt, L, r.stream));
                LEAF(asd_transpose_f16(d_ao + ro, L, C, C, tA, L, r.stream));
                gemm(r, pt, L, L, tA, C, L, L, dv + ro, C);                                  // dV = P^T dO
                gemm(r, d_ao + ro, L, C, sv.v + ro, L, C, C, dp, L);                          // dP = dO V^T
                LEAF(asd_softmax_bwd_f16(P, dp, L, L, L, 1.0f / sqrtf((float)C), dsm, r.stream));
                LEAF(asd_transpose_f16(sv.k + ro, L, C, C, tA, L, r.stream));
                gemm(r, dsm, L, L, tA, C, L, L, dq + ro, C);                                  // dQ = dS K
                LEAF(asd_transpose_f16(dsm, L, L, L, pt, L, r.stream));
                LEAF(asd_transpose_f16(sv.q + ro, L, C, C, tA, L, r.stream));
                gemm(r, pt, L, L, tA, C, L, L, dk + ro, C);                                   // dK = dS^T Q
            }
            half_t* d1 = other(2);
            gemm(r, dq, M, C, n.w(p + ".q.wt"), C, C, C, d1, C);
            half_t* d2 = other(3);
            { GemmOpt o; o.residual = d1; o.ldr = C; gemm(r, dk, M, C, n.w(p + ".k.wt"), C, C, C, d2, C, o); }
            Act a1;
            { GemmOpt o = feeds_gn_bwd(&a1, hw, sv.x, sv.st1, n.w(p + ".norm.weight"), n.w(p + ".norm.bias"), 0);
              o.residual = d2; o.ldr = C; gemm(r, dv, M, C, n.w(p + ".v.wt"), C, C, C, d1, C, o); }
            half_t* dx = other(3);
            gn_bwd(r, sv.x, d1, C, B, hw, n.w(p + ".norm.weight"), n.w(p + ".norm.bias"), 0, sv.st1, dy, dx, &a1);   // + the residual path
            cur = (cur + 3) & 3;
            dy = buf[cur];
        } else if (l.kind == 2) {
            half_t* dx = other(1);
            // (hh, ww even: the low-resolution image is exactly half; 9 of the 16 tap slots carry weights)
            conv3x3(r, dy, B, hh / 2, ww / 2, l.cout, n.w(l.name + ".bwd"), l.cin, dx, hh, ww, 1, 1, 3);
            cur = (cur + 1) & 3;
            dy = buf[cur];
        } else if (l.kind == 1) {
            const std::string& p = l.name;
            half_t* d3 = other(1);
            Act a3, a1;
            conv3x3(r, dy, B, hh, ww, l.cout, n.w(p + ".conv2.bwd"), l.cout, d3, hh, ww, 1, 1, 0,
                    feeds_gn_bwd(&a3, hw, sv.t2, sv.st2, n.w(p + ".norm2.weight"), n.w(p + ".norm2.bias"), 1));
            half_t* d2 = other(2);
            gn_bwd(r, sv.t2, d3, l.cout, B, hw, n.w(p + ".norm2.weight"), n.w(p + ".norm2.bias"), 1, sv.st2, nullptr, d2, &a3);
            half_t* d1 = other(1);
            conv3x3(r, d2, B, hh, ww, l.cout, n.w(p + ".conv1.bwd"), l.cin, d1, hh, ww, 1, 1, 0,
                    feeds_gn_bwd(&a1, hw, sv.x, sv.st1, n.w(p + ".norm1.weight"), n.w(p + ".norm1.bias"), 1));
            if (n.has(p + ".nin.w")) {
                half_t* dmain = other(2);
                gn_bwd(r, sv.x, d1, l.cin, B, hw, n.w(p + ".norm1.weight"), n.w(p + ".norm1.bias"), 1, sv.st1, nullptr, dmain, &a1);
                half_t* dx = other(3);
                GemmOpt o; o.residual = dmain; o.ldr = l.cin;
                gemm(r, dy, M, l.cout, n.w(p + ".nin.wt"), l.cin, l.cout, l.cout, dx, l.cin, o);
                cur = (cur + 3) & 3;
            } else {
                half_t* dx = other(2);
                gn_bwd(r, sv.x, d1, l.cin, B, hw, n.w(p + ".norm1.weight"), n.w(p + ".norm1.bias"), 1, sv.st1, dy, dx, &a1);   // + shortcut
                cur = (cur + 2) & 3;
            }
            dy = buf[cur];
        } else {
            conv3x3(r, dy, B, hh, ww, l.cout, n.w(l.name + ".bwd"), pad32(l.cin), dx32, hh, ww, 1, 1, 0);
        }
    }
}

int make_zero_page(Net& n) {
    if (hipMalloc(&n.zero_page, 256) != hipSuccess || hipMemset(n.zero_page, 0, 256) != hipSuccess) {
        asd_set_error("zero page allocation failed");
        return ASD_ERR_LAUNCH;
    }
    return ASD_OK;
}

// (the zero page — source of out-of-range taps — is the only device memory a handle owns; it is created with the first binding, so that
// create / weight_info work without a device)
int bind(Net& n, const void* const* ptrs, int count) {
    if (count != (int)n.specs.size()) { asd_set_error("expected %d weight pointers, got %d", (int)n.specs.size(), count); return ASD_ERR_ARG; }
    for (int i = 0; i < count; ++i) {
        if (!ptrs[i]) { asd_set_error("weight %s is NULL", n.specs[i].name.c_str()); return ASD_ERR_ARG; }
        n.ptr[i] = ptrs[i];
    }
    if (!n.zero_page && make_zero_page(n) != ASD_OK) return ASD_ERR_LAUNCH;
    n.bound = true;
    return ASD_OK;
}

const size_t TUNE_SCRATCH = (size_t)512 << 20;

// workspace layout: [split-K scratch][activations]
template <class F>
size_t plan_bytes(F&& pass, bool tune) {
    Run r;
    r.dry = true;
    r.tune = tune;        // sizing rule of the statistics records (upper bound while plans may still change)
    pass(r);
    const size_t scratch = tune ? TUNE_SCRATCH : r.scratch_need;
    return ((scratch + 255) & ~(size_t)255) + r.mem.off + 256;
}
// *scratch_io: in = scratch size to use (when `given`), out = the size used — the VAE backward must lay the workspace out exactly
// as its forward did, whatever happened to the plan table in between
// checked: the caller has already run this very pass on this workspace size under the current plan table (asd_gemm_plan_generation) and
// passes its scratch size — the sizing walk over the network is skipped (two of them took ~100 us of host time in front of the first
// launch of every VAE pass, with the GPU idle: the host is not ahead right after the renderer's read-back)
template <class F>
int run_pass(F&& pass, void* workspace, size_t workspace_bytes, hipStream_t stream, bool tune, const void* zero_page,
             size_t* scratch_io = nullptr, bool given = false, bool checked = false) {
    size_t scratch = 0;
    if (checked && given) scratch = *scratch_io;
    else {
        Run dry;
        dry.dry = true;
        dry.tune = tune;
        pass(dry);
        scratch = ((tune ? TUNE_SCRATCH : dry.scratch_need) + 255) & ~(size_t)255;
        if (given) scratch = *scratch_io;
        if (scratch_io) *scratch_io = scratch;
        if (scratch + dry.mem.off > workspace_bytes) {
            asd_set_error("workspace of %zu bytes is too small (need %zu)", workspace_bytes, scratch + dry.mem.off);
            return ASD_ERR_ARG;
        }
    }
    Run r;
    r.dry = false; r.tune = tune; r.stream = stream; r.zero_page = zero_page;
    r.scratch = (float*)workspace; r.scratch_bytes = scratch;
    r.mem.base = (char*)workspace + scratch;
    pass(r);
    return r.status;
}

}  // namespace

extern "C" {

// ---- UNet ----------------------------------------------------------------------------------------------------------------
int asd_unet_create(const asd_unet_desc* desc, asd_unet** out) {
    ASD_CHECK_ARG(desc && out, "null argument");
    ASD_CHECK_ARG(desc->n_levels >= 1 && desc->n_levels <= 8 && desc->model_channels % 32 == 0 && desc->transformer_depth >= 1, "bad descriptor");
    ASD_CHECK_ARG(desc->num_head_channels == 64, "the attention kernel is built for head_dim 64");
    UNet* n = new UNet();
    n->d = *desc;
    unet_build(*n);
    *out = (asd_unet*)n;
    return ASD_OK;
}
void asd_unet_destroy(asd_unet* h) {
    UNet* n = (UNet*)h;
    if (!n) return;
    if (n->zero_page) (void)hipFree(n->zero_page);
    delete n;
}
int32_t asd_unet_num_weights(const asd_unet* h) { return h ? (int32_t)((const UNet*)h)->specs.size() : 0; }
int asd_unet_weight_info(const asd_unet* h, int32_t i, asd_weight_info* info) {
    const UNet* n = (const UNet*)h;
    ASD_CHECK_ARG(n && info && i >= 0 && i < (int)n->specs.size(), "bad argument");
    info->name = n->specs[i].name.c_str(); info->rows = n->specs[i].rows; info->cols = n->specs[i].cols;
    return ASD_OK;
}
int asd_unet_bind_weights(asd_unet* h, const void* const* ptrs, int32_t count) {
    ASD_CHECK_ARG(h && ptrs, "null argument");
    return bind(*(UNet*)h, ptrs, count);
}
int64_t asd_unet_workspace_bytes_shared(asd_unet* h, int32_t batch, int32_t H, int32_t W, int32_t n_ctx, int32_t num_frames, int32_t tune,
                                        int32_t n_uniq) {
    UNet* n = (UNet*)h;
    if (!n || batch < 1 || H < 1 || W < 1 || n_ctx < 1) return -1;
    return (int64_t)plan_bytes([&](Run& r) { unet_run(*n, r, nullptr, nullptr, nullptr, nullptr, batch, H, W, n_ctx, num_frames, nullptr,
                                                      nullptr, nullptr, n_uniq); }, tune != 0);
}
int64_t asd_unet_workspace_bytes(asd_unet* h, int32_t batch, int32_t H, int32_t W, int32_t n_ctx, int32_t num_frames, int32_t tune) {
    return asd_unet_workspace_bytes_shared(h, batch, H, W, n_ctx, num_frames, tune, 0);
}
int asd_unet_fwd(asd_unet* h, const void* x_nhwc, const float* t, const void* context, const void* camera, int32_t batch, int32_t H,
                 int32_t W, int32_t n_ctx, int32_t num_frames, void* workspace, int64_t workspace_bytes, float* eps_nhwc, int32_t tune,
                 void* stream) {
    return asd_unet_fwd_shared(h, x_nhwc, t, context, camera, batch, H, W, n_ctx, num_frames, nullptr, nullptr, 0, workspace, workspace_bytes,
                               eps_nhwc, tune, stream);
}

int asd_unet_fwd_shared(asd_unet* h, const void* x_nhwc, const float* t, const void* context, const void* camera, int32_t batch, int32_t H,
                        int32_t W, int32_t n_ctx, int32_t num_frames, const int32_t* uniq_src_dev, const int32_t* expand_dev, int32_t n_uniq,
                        void* workspace, int64_t workspace_bytes, float* eps_nhwc, int32_t tune, void* stream) {
    UNet* n = (UNet*)h;
    if (!uniq_src_dev || !expand_dev || n_uniq >= batch) n_uniq = 0;
    ASD_CHECK_ARG(n_uniq == 0 || (n_uniq > 0 && n_uniq % (num_frames > 0 ? num_frames : 1) == 0), "distinct inputs must come in whole groups of num_frames");
    ASD_CHECK_ARG(n && x_nhwc && t && context && workspace && eps_nhwc, "null argument");
    ASD_CHECK_ARG(n->bound, "weights are not bound (asd_unet_bind_weights)");
    ASD_CHECK_ARG((n->d.camera_dim > 0) == (camera != nullptr), "camera is given iff the UNet is camera-conditioned");
    ASD_CHECK_ARG(num_frames >= 1 && batch % num_frames == 0, "[UNet] input batch size must be dividable by num_frames!");
    const int levels = n->d.n_levels - 1;
    ASD_CHECK_ARG(H % (1 << levels) == 0 && W % (1 << levels) == 0, "H and W must be divisible by 2^(levels-1)");
    return run_pass([&](Run& r) { unet_run(*n, r, (const half_t*)x_nhwc, t, (const half_t*)context, (const half_t*)camera, batch, H, W, n_ctx,
                                           num_frames, eps_nhwc, (const int*)uniq_src_dev, (const int*)expand_dev, n_uniq); },
                    workspace, (size_t)workspace_bytes, (hipStream_t)stream, tune != 0, n->zero_page);
}

// ---- VAE encoder -----------------------------------------------------------------------------------------------------------
int asd_vae_enc_create(const asd_vae_desc* desc, asd_vae_enc** out) {
    ASD_CHECK_ARG(desc && out, "null argument");
    ASD_CHECK_ARG(desc->n_levels >= 1 && desc->n_levels <= 8 && desc->ch % 32 == 0 && (2 * desc->embed_dim) % 4 == 0, "bad descriptor");
    Vae* n = new Vae();
    n->d = *desc;
    vae_build(*n);
    *out = (asd_vae_enc*)n;
    return ASD_OK;
}
void asd_vae_enc_destroy(asd_vae_enc* h) {
    Vae* n = (Vae*)h;
    if (!n) return;
    if (n->zero_page) (void)hipFree(n->zero_page);
    delete n;
}
int32_t asd_vae_enc_num_weights(const asd_vae_enc* h) { return h ? (int32_t)((const Vae*)h)->specs.size() : 0; }
int asd_vae_enc_weight_info(const asd_vae_enc* h, int32_t i, asd_weight_info* info) {
    const Vae* n = (const Vae*)h;
    ASD_CHECK_ARG(n && info && i >= 0 && i < (int)n->specs.size(), "bad argument");
    info->name = n->specs[i].name.c_str(); info->rows = n->specs[i].rows; info->cols = n->specs[i].cols;
    return ASD_OK;
}
int asd_vae_enc_bind_weights(asd_vae_enc* h, const void* const* ptrs, int32_t count) {
    ASD_CHECK_ARG(h && ptrs, "null argument");
    return bind(*(Vae*)h, ptrs, count);
}
// one workspace serves a forward and the backward that follows it: [scratch][forward activations][backward temporaries]
int64_t asd_vae_enc_workspace_bytes(asd_vae_enc* h, int32_t batch, int32_t H, int32_t W, int32_t tune) {
    Vae* n = (Vae*)h;
    if (!n || batch < 1 || H < 8 || W < 8) return -1;
    {   // the answer of the last forward pass, while shape and plans are the same
        const Vae::Sized& z = n->sized;
        if (z.gen != 0 && z.gen == asd_gemm_plan_generation() && z.batch == batch && z.H == H && z.W == W && z.tune == tune) return z.need;
    }
    return (int64_t)plan_bytes([&](Run& r) {
        std::vector<VSaved> saved;
        vae_forward(*n, r, nullptr, batch, H, W, nullptr, saved, true);
        vae_backward(*n, r, saved, nullptr, batch, nullptr);
    }, tune != 0);
}
int asd_vae_enc_fwd(asd_vae_enc* h, const void* x_nhwc32, int32_t batch, int32_t H, int32_t W, void* workspace, int64_t workspace_bytes,
                    float* moments_nhwc, int32_t tune, void* stream) {
    Vae* n = (Vae*)h;
    ASD_CHECK_ARG(n && x_nhwc32 && workspace && moments_nhwc, "null argument");
    ASD_CHECK_ARG(n->bound, "weights are not bound (asd_vae_enc_bind_weights)");
    ASD_CHECK_ARG(H > 0 && W > 0 && H % (1 << (n->d.n_levels - 1)) == 0 && W % (1 << (n->d.n_levels - 1)) == 0,
                  "H and W must be multiples of 2^(levels - 1): every Downsample halves the image exactly (parity-form input gradient)");
    // size check against the forward + backward plan, so that the backward can never overrun what the forward accepted
    const uint64_t gen = asd_gemm_plan_generation();
    Vae::Sized& z = n->sized;
    static const bool cache_on = !(getenv("ASD_VAE_SIZING_CACHE") && getenv("ASD_VAE_SIZING_CACHE")[0] == '0');       // A/B switch (tools)
    const bool known = cache_on && z.gen == gen && z.batch == batch && z.H == H && z.W == W && z.tune == tune && workspace_bytes >= z.need;
    const int64_t need = known ? z.need : asd_vae_enc_workspace_bytes(h, batch, H, W, tune);
    if (need < 0 || need > workspace_bytes) { asd_set_error("workspace of %lld bytes is too small (need %lld)", (long long)workspace_bytes, (long long)need); return ASD_ERR_ARG; }
    // the scratch region is sized for both passes (the dry pass walks the backward as well) and remembered per workspace
    size_t scratch = known ? z.scratch : 0;
    std::vector<int> log;
    const int st = run_pass([&](Run& r) {
        std::vector<VSaved> saved;
        if (!r.dry) r.rec_log = &log;
        vae_forward(*n, r, (const half_t*)x_nhwc32, batch, H, W, moments_nhwc, saved, true);
        if (r.dry) vae_backward(*n, r, saved, nullptr, batch, nullptr);
    }, workspace, (size_t)workspace_bytes, (hipStream_t)stream, tune != 0, n->zero_page, &scratch, known, known);
    if (st == ASD_OK) {
        n->scratch_of[workspace] = scratch; n->recs_of[workspace] = log;
        const uint64_t after = asd_gemm_plan_generation();          // tuning inside the pass changes the plans: nothing to remember then
        n->gen_of[workspace] = after == gen ? gen : 0;
        if (after == gen) { z.batch = batch; z.H = H; z.W = W; z.tune = tune; z.gen = gen; z.need = need; z.scratch = scratch; }
        else z.gen = 0;
    }
    return st;
}
int asd_vae_enc_bwd(asd_vae_enc* h, const float* d_moments_nhwc, int32_t batch, int32_t H, int32_t W, void* workspace, int64_t workspace_bytes,
                    void* dx_nhwc32, int32_t tune, void* stream) {
    Vae* n = (Vae*)h;
    ASD_CHECK_ARG(n && d_moments_nhwc && workspace && dx_nhwc32, "null argument");
    ASD_CHECK_ARG(n->bound, "weights are not bound (asd_vae_enc_bind_weights)");
    auto it = n->scratch_of.find(workspace);
    ASD_CHECK_ARG(it != n->scratch_of.end(), "no forward pass has run on this workspace");
    size_t scratch = it->second;
    const std::vector<int>& log = n->recs_of[workspace];
    return run_pass([&](Run& r) {
        std::vector<VSaved> saved;
        r.rec_replay = &log;
        r.rec_pos = 0;
        vae_forward(*n, r, nullptr, batch, H, W, nullptr, saved, false);     // replays the forward's allocation sequence: same addresses
        r.rec_replay = nullptr;                                              // the backward's own launches size their records from the plans
        vae_backward(*n, r, saved, d_moments_nhwc, batch, (half_t*)dx_nhwc32);
    }, workspace, (size_t)workspace_bytes, (hipStream_t)stream, tune != 0, n->zero_page, &scratch, true,
       /* checked: the forward on this workspace sized both passes under the same plans and shape */
       n->gen_of[workspace] != 0 && n->gen_of[workspace] == asd_gemm_plan_generation() && n->sized.gen == n->gen_of[workspace] &&
           n->sized.batch == batch && n->sized.H == H && n->sized.W == W && n->sized.tune == tune && workspace_bytes >= n->sized.need);
}

}  // extern "C"
