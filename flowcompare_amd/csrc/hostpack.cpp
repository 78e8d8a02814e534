// Host-side helpers: weight table, double-precision folding, packing, device arena.
#include "hostpack.h"

#include <algorithm>
#include <cstring>
#include <cstdio>

namespace fc {

static thread_local std::string g_last_error;
void set_last_error(const std::string& m) { g_last_error = m; }
const char* get_last_error() { return g_last_error.c_str(); }

// ---------------------------------------------------------------- profiler
namespace {
struct ProfRec { hipEvent_t a, b; int name; double flops, bytes; };
struct ProfState {
    bool on = false;
    std::string filter;               // non-empty: only launches whose kernel name contains it are bracketed
    int stride = 1, seen = 0;         // of the launches that pass the filter every stride-th one is bracketed (fc_profile_stride)
    std::vector<std::string> names;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    int name_id(const char* n) {
        for (size_t i = 0; i < names.size(); ++i) if (names[i] == n) return (int)i;
        names.push_back(n);
        return (int)names.size() - 1;
    }
    hipEvent_t ev() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) throw Error(FC_ERR_HIP, "hipEventCreate failed");
        return e;
    }
};
ProfState g_prof;
}  // namespace
bool prof_enabled() { return g_prof.on; }
ProfScope::ProfScope(const char* name, double flops, double bytes, hipStream_t stream) : on(g_prof.on), s(stream) {
    if (on && !g_prof.filter.empty() && !strstr(name, g_prof.filter.c_str())) on = false;
    if (on && g_prof.stride > 1 && (g_prof.seen++ % g_prof.stride) != 0) on = false;
    if (!on) return;
    ProfRec r{g_prof.ev(), g_prof.ev(), g_prof.name_id(name), flops, bytes};
    (void)hipEventRecord(r.a, s);
    g_prof.recs.push_back(r);
    slot = (int)g_prof.recs.size() - 1;
}
ProfScope::~ProfScope() {
    if (on && slot >= 0) (void)hipEventRecord(g_prof.recs[slot].b, s);
}
void prof_set(bool on) { g_prof.on = on; }
void prof_filter(const char* substr) { g_prof.filter = substr ? substr : ""; }
void prof_stride(int n) { g_prof.stride = n > 1 ? n : 1; g_prof.seen = 0; }
void prof_reset() {
    for (auto& r : g_prof.recs) { g_prof.pool.push_back(r.a); g_prof.pool.push_back(r.b); }
    g_prof.recs.clear();
}
std::string prof_report_json() {
    struct Agg { long long n = 0; double ms = 0, flops = 0, bytes = 0; };
    std::vector<Agg> agg(g_prof.names.size());
    for (auto& r : g_prof.recs) {
        FC_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        FC_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        Agg& a = agg[r.name];
        a.n += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
    }
    std::string out = "[";
    bool first = true;
    for (size_t i = 0; i < agg.size(); ++i) {
        if (!agg[i].n) continue;
        char buf[512];
        snprintf(buf, sizeof buf, "%s{\"kernel\": \"%s\", \"launches\": %lld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
                 first ? "" : ", ", g_prof.names[i].c_str(), agg[i].n, agg[i].ms, agg[i].flops, agg[i].bytes);
        out += buf;
        first = false;
    }
    return out + "]";
}

// ---------------------------------------------------------------- arena
void* DeviceArena::alloc_bytes(size_t n) {
    constexpr size_t CHUNK = 64u << 20;
    n = round_up_sz(std::max<size_t>(n, 1), 256);
    std::lock_guard<std::mutex> lock(mu);
    total += n;
    if (n > CHUNK / 4) {                                  // large tensors get their own allocation
        void* p = nullptr;
        FC_HIP(hipMalloc(&p, n));
        blocks.push_back(p);
        return p;
    }
    if (n > cur_left) {
        void* p = nullptr;
        FC_HIP(hipMalloc(&p, CHUNK));
        blocks.push_back(p);
        cur = (char*)p;
        cur_left = CHUNK;
    }
    void* out = cur;
    cur += n;
    cur_left -= n;
    return out;
}
float* DeviceArena::upload(const std::vector<float>& host) {
    float* d = alloc_floats(host.size());
    if (!host.empty()) FC_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    return d;
}
DeviceArena::~DeviceArena() {
    for (void* p : blocks) (void)hipFree(p);
}

// ---------------------------------------------------------------- weight table
int64_t HostTensor::numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
}
WeightTable::WeightTable(const fc_tensor* tensors, int n) {
    if (n < 0 || (n > 0 && !tensors)) throw Error(FC_ERR_INVALID, "null tensor list");
    for (int i = 0; i < n; ++i) {
        const fc_tensor& ft = tensors[i];
        if (!ft.name || ft.ndim < 0 || ft.ndim > 4) throw Error(FC_ERR_INVALID, "malformed fc_tensor");
        HostTensor h;
        h.data = ft.data;
        for (int d = 0; d < ft.ndim; ++d) h.shape.push_back(ft.shape[d]);
        if (h.numel() > 0 && !ft.data) throw Error(FC_ERR_INVALID, std::string("tensor without data: ") + ft.name);
        t[ft.name] = h;
    }
}
const HostTensor& WeightTable::get(const std::string& name) const {
    auto it = t.find(name);
    if (it == t.end()) throw Error(FC_ERR_MISSING, "state_dict entry missing: " + name);
    return it->second;
}
const HostTensor& WeightTable::get(const std::string& name, std::initializer_list<int64_t> shape) const {
    const HostTensor& h = get(name);
    std::vector<int64_t> want(shape);
    if (h.shape != want) {
        std::string s = "state_dict entry " + name + " has shape [";
        for (auto d : h.shape) s += std::to_string(d) + ",";
        s += "] expected [";
        for (auto d : want) s += std::to_string(d) + ",";
        throw Error(FC_ERR_SHAPE, s + "]");
    }
    return h;
}

// ---------------------------------------------------------------- dense double helpers
MatD mat_from(const HostTensor& t) {
    if (t.shape.size() < 2) throw Error(FC_ERR_SHAPE, "expected a matrix");
    int64_t k = 1;
    for (size_t d = 1; d < t.shape.size(); ++d) k *= t.shape[d];
    MatD m((int)t.shape[0], (int)k);
    for (size_t i = 0; i < m.v.size(); ++i) m.v[i] = t.data[i];
    return m;
}
VecD vec_from(const HostTensor& t) {
    VecD v((size_t)t.numel());
    for (size_t i = 0; i < v.size(); ++i) v[i] = t.data[i];
    return v;
}
MatD matmul(const MatD& a, const MatD& b) {
    if (a.cols != b.rows) throw Error(FC_ERR_SHAPE, "matmul shape mismatch");
    MatD c(a.rows, b.cols);
    for (int i = 0; i < a.rows; ++i) {
        double* ci = &c.v[(size_t)i * c.cols];
        for (int k = 0; k < a.cols; ++k) {
            const double aik = a.at(i, k);
            if (aik == 0.0) continue;
            const double* bk = &b.v[(size_t)k * b.cols];
            for (int j = 0; j < b.cols; ++j) ci[j] += aik * bk[j];
        }
    }
    return c;
}
MatD expm_double(const MatD& w) {
    const int n = w.rows;
    double nrm = 0;
    for (int i = 0; i < n; ++i) {
        double r = 0;
        for (int j = 0; j < n; ++j) r += std::fabs(w.at(i, j));
        nrm = std::max(nrm, r);
    }
    int sq = 0;
    while (nrm > 0.25) { nrm *= 0.5; ++sq; }
    MatD x = w;
    const double sc = std::ldexp(1.0, -sq);
    for (auto& e : x.v) e *= sc;
    MatD s(n, n), term(n, n);
    for (int i = 0; i < n; ++i) { s.at(i, i) = 1.0; term.at(i, i) = 1.0; }
    for (int k = 1; k <= 24; ++k) {
        term = matmul(term, x);
        for (auto& e : term.v) e /= k;
        for (size_t i = 0; i < s.v.size(); ++i) s.v[i] += term.v[i];
    }
    for (int i = 0; i < sq; ++i) s = matmul(s, s);
    return s;
}
double slogdet_abs(const MatD& w) {
    const int n = w.rows;
    MatD a = w;
    double ld = 0;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(a.at(r, c)) > std::fabs(a.at(piv, c))) piv = r;
        if (a.at(piv, c) == 0.0) throw Error(FC_ERR_INVALID, "singular permuter matrix");
        if (piv != c) for (int j = 0; j < n; ++j) std::swap(a.at(piv, j), a.at(c, j));
        ld += std::log(std::fabs(a.at(c, c)));
        for (int r = c + 1; r < n; ++r) {
            const double f = a.at(r, c) / a.at(c, c);
            if (f != 0.0) for (int j = c; j < n; ++j) a.at(r, j) -= f * a.at(c, j);
        }
    }
    return ld;
}
MatD inverse_double(const MatD& w) {
    const int n = w.rows;
    MatD a = w, inv(n, n);
    for (int i = 0; i < n; ++i) inv.at(i, i) = 1.0;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(a.at(r, c)) > std::fabs(a.at(piv, c))) piv = r;
        if (a.at(piv, c) == 0.0) throw Error(FC_ERR_INVALID, "singular matrix in inverse");
        if (piv != c) for (int j = 0; j < n; ++j) { std::swap(a.at(piv, j), a.at(c, j)); std::swap(inv.at(piv, j), inv.at(c, j)); }
        const double d = 1.0 / a.at(c, c);
        for (int j = 0; j < n; ++j) { a.at(c, j) *= d; inv.at(c, j) *= d; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = a.at(r, c);
            if (f == 0.0) continue;
            for (int j = 0; j < n; ++j) { a.at(r, j) -= f * a.at(c, j); inv.at(r, j) -= f * inv.at(c, j); }
        }
    }
    return inv;
}

// ---------------------------------------------------------------- bf16 limbs
static inline unsigned short f32_to_bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)((u >> 16) | ((u & 0xffffu) ? 0x40u : 0u));   // inf / nan
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf16_to_f32(unsigned short h) {
    uint32_t u = (uint32_t)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
bool g_pack_bf16_limbs = true;

std::vector<unsigned short> make_bf16_limbs(const std::vector<float>& w, int n_alloc, int K_pad) {
    const int kt16 = K_pad / 16;
    std::vector<unsigned short> w3((size_t)n_alloc * kt16 * 48, 0);
    for (int n = 0; n < n_alloc; ++n)
        for (int k = 0; k < K_pad; ++k) {
            const float x = w[(size_t)n * K_pad + k];
            if (x == 0.f) continue;
            const unsigned short h = f32_to_bf16_rne(x);
            const float r1 = x - bf16_to_f32(h);
            const unsigned short m = f32_to_bf16_rne(r1);
            const float r2 = r1 - bf16_to_f32(m);
            const unsigned short l = f32_to_bf16_rne(r2);
            unsigned short* dst = &w3[((size_t)n * kt16 + k / 16) * 48 + k % 16];
            dst[0] = h; dst[16] = m; dst[32] = l;
        }
    return w3;
}

// fp16 limb image for the split-fp16 GEMM loop: [n_alloc][K_pad/16][2][16] = hi, lo' with w ~= hi + lo'/2048 (gemm.hip).
// Returns an empty vector when an entry does not fit fp16 (|w| >= 65504 or not finite): that layer keeps the bf16 limbs only.
std::vector<unsigned short> make_f16_limbs(const std::vector<float>& w, int n_alloc, int K_pad) {
    const int kt16 = K_pad / 16;
    std::vector<unsigned short> w2((size_t)n_alloc * kt16 * 32, 0);
    for (int n = 0; n < n_alloc; ++n)
        for (int k = 0; k < K_pad; ++k) {
            const float x = w[(size_t)n * K_pad + k];
            if (x == 0.f) continue;
            if (!(std::fabs(x) < 65504.0f)) return {};
            const _Float16 h = (_Float16)x;                                   // round to nearest even
            const _Float16 l = (_Float16)((x - (float)h) * 2048.0f);
            unsigned short hb, lb;
            std::memcpy(&hb, &h, 2); std::memcpy(&lb, &l, 2);
            unsigned short* dst = &w2[((size_t)n * kt16 + k / 16) * 32 + k % 16];
            dst[0] = hb; dst[16] = lb;
        }
    return w2;
}

// ---------------------------------------------------------------- index maps
std::vector<int> map_prefix(int n_src, int n_pad) {
    std::vector<int> m(n_pad, -1);
    for (int i = 0; i < n_src && i < n_pad; ++i) m[i] = i;
    return m;
}
std::vector<int> map_xlayout(int d1, int d1_pad, int d2, int d2_pad) {
    std::vector<int> m(d1_pad + d2_pad, -1);
    for (int i = 0; i < d1; ++i) m[i] = i;
    for (int j = 0; j < d2; ++j) m[d1_pad + j] = d1 + j;
    return m;
}
std::vector<int> map_pairs(int d2, int second_half_offset) {
    const int np = (d2 + 31) / 32;
    std::vector<int> m(np * 64, -1);
    for (int p = 0; p < np; ++p)
        for (int c = 0; c < 32; ++c) {
            const int j = 32 * p + c;
            if (j < d2) { m[64 * p + c] = j; m[64 * p + 32 + c] = second_half_offset + j; }
        }
    return m;
}
std::vector<int> map_concat(const std::vector<std::vector<int>>& parts) {
    std::vector<int> m;
    for (auto& p : parts) m.insert(m.end(), p.begin(), p.end());
    return m;
}

PackedLinear pack_linear(DeviceArena& arena, const MatD& W, const VecD& bias, const VecD& colvec, const std::vector<int>& nmap,
                         const std::vector<int>& kmap, const std::vector<int>& seg_k) {
    PackedLinear L;
    L.N_pad = (int)nmap.size();
    L.K_pad = (int)kmap.size();
    if (L.N_pad % 32 || L.K_pad % 32) throw Error(FC_ERR_INVALID, "pack_linear: maps must be padded to 32");
    int ks = 0;
    L.nseg = (int)seg_k.size();
    if (L.nseg < 1 || L.nseg > 3) throw Error(FC_ERR_INVALID, "pack_linear: 1..3 segments");
    for (int i = 0; i < L.nseg; ++i) { L.seg_k[i] = seg_k[i]; ks += seg_k[i]; }
    if (ks != L.K_pad) throw Error(FC_ERR_INVALID, "pack_linear: segments do not cover K");
    L.n_alloc = gemm_n_alloc(L.N_pad);
    std::vector<float> w((size_t)L.n_alloc * L.K_pad, 0.f), b(L.n_alloc, 0.f), cv(L.n_alloc, 0.f);
    for (int n = 0; n < L.N_pad; ++n) {
        const int sn = nmap[n];
        if (sn < 0) continue;
        if (sn >= W.rows) throw Error(FC_ERR_SHAPE, "pack_linear: row map out of range");
        float* wr = &w[(size_t)n * L.K_pad];
        for (int k = 0; k < L.K_pad; ++k) {
            const int sk = kmap[k];
            if (sk >= 0) {
                if (sk >= W.cols) throw Error(FC_ERR_SHAPE, "pack_linear: column map out of range");
                wr[k] = (float)W.at(sn, sk);
            }
        }
        if (!bias.empty()) b[n] = (float)bias[sn];
        if (!colvec.empty()) cv[n] = (float)colvec[sn];
    }
    for (int n : nmap) L.n_true += n >= 0;
    for (int k : kmap) L.k_true += k >= 0;
    for (float x : w) L.wmax = std::max(L.wmax, std::fabs(x));
    L.W = arena.upload(w);
    L.bias = arena.upload(b);
    if (g_pack_bf16_limbs && L.K_pad % 16 == 0) {
        // the limb images are made ON THE DEVICE from the fp32 matrix just uploaded (same roundings as make_*_limbs above, which remain
        // as the host statement of the layout): the host neither converts nor uploads 10 of the 14 bytes per weight.  A matrix with an
        // entry outside fp16's range gets no fp16 image (the split-fp16 loop then never runs on it).
        bool fits16 = true;
        for (float x : w) if (!(std::fabs(x) < 65504.0f)) { fits16 = false; break; }
        const size_t n16 = (size_t)L.n_alloc * (L.K_pad / 16);
        L.W3 = (unsigned short*)arena.alloc_bytes(n16 * 48 * sizeof(unsigned short));
        L.W2 = fits16 ? (unsigned short*)arena.alloc_bytes(n16 * 32 * sizeof(unsigned short)) : nullptr;
        launch_limb_images(L.W, L.n_alloc, L.K_pad, L.W3, L.W2, nullptr);
    }
    L.colvec = colvec.empty() ? nullptr : arena.upload(cv);
    return L;
}

void pack_mlp_mid(DeviceArena& arena, const WeightTable& wt, const std::string& prefix, PackedMLP& out) {
    out.sizes.clear();
    out.mid.clear();
    const HostTensor& w_in = wt.get(prefix + ".in_layer.weight");
    if (w_in.shape.size() != 2) throw Error(FC_ERR_SHAPE, prefix + ".in_layer.weight must be 2-D");
    out.sizes.push_back((int)w_in.shape[0]);
    for (int i = 0;; ++i) {
        const std::string n = prefix + ".layers." + std::to_string(i);
        if (!wt.has(n + ".weight")) break;
        const HostTensor& w = wt.get(n + ".weight");
        if (w.shape.size() != 2 || w.shape[1] != out.sizes.back()) throw Error(FC_ERR_SHAPE, n + ".weight: input width mismatch");
        const int no = (int)w.shape[0], ni = (int)w.shape[1];
        if (i % 2 == 1 && no != out.sizes[out.sizes.size() - 2])
            throw Error(FC_ERR_SHAPE, n + ": residual add needs matching widths (models/nets.py:27)");
        wt.get(n + ".bias", {no});
        out.mid.push_back(pack_linear(arena, mat_from(w), vec_from(wt.get(n + ".bias")), {}, map_prefix(no, round_up(no, 32)),
                                      map_prefix(ni, round_up(ni, 32)), {round_up(ni, 32)}));
        out.sizes.push_back(no);
    }
    const HostTensor& w_out = wt.get(prefix + ".out_layer.weight");
    if (w_out.shape.size() != 2 || w_out.shape[1] != out.sizes.back()) throw Error(FC_ERR_SHAPE, prefix + ".out_layer.weight: input width mismatch");
}
int run_mlp_hidden_generic(const PackedMLP& m, const ASeg* in_segs, const float* rowscal, int act, float* const h[3], int ldh, int rows,
                           hipStream_t s, int rows_valid, unsigned short* last_limbs, float last_scale, bool wide) {
    // Full limb chain (with `last_limbs`, inside a guard scope, every layer with an fp16 image and 128-multiple widths): EVERY hidden
    // activation exists only as the fp16 limb image its producer's epilogue writes (same 4 bytes per element as fp32, in the h[] buffers);
    // the consumers copy it (A16: no conversion in the main loop, LDS-DMA) and the odd layers read their residual from the image too.
    bool all = last_limbs && gemm_limb_chain_all_ok() && m.in_layer.W2 && m.in_layer.N_pad % 128 == 0 && m.in_layer.N_pad <= ldh;
    // wide: the hidden layers on the 256 x 256 one-accumulator kernel (spline_wide.hip), every intermediate image in the one-accumulator form
    bool wide_ok = wide && act == FC_ACT_GELU && rows % 256 == 0 && !m.mid.empty();
    for (const PackedLinear& L : m.mid) wide_ok = wide_ok && L.W1 && !L.w1_permuted && L.N_pad % 256 == 0 && L.K_pad % 64 == 0;
    int prev_n = m.in_layer.N_pad;
    for (const PackedLinear& L : m.mid) { all = all && L.W2 && L.nseg == 1 && L.N_pad % 128 == 0 && L.K_pad == prev_n && L.N_pad <= ldh; prev_n = L.N_pad; }
    GemmEpi e{};
    e.act = act; e.C = h[0]; e.ldc = ldh; e.rowscal = rowscal; e.rows_valid = rows_valid;
    if (last_limbs && m.mid.empty()) { e.C = nullptr; e.C16 = last_limbs; e.c16_scale = last_scale; }
    else if (all) { e.C = nullptr; e.C16 = reinterpret_cast<unsigned short*>(h[0]); }
    wide_ok = wide_ok && all;
    if (wide_ok) e.c16_scale = kOneAccActScale;
    launch_gemm(m.in_layer, in_segs, rows, e, EPI_LINEAR, s);
    int cur = 0, keep = -1;
    for (size_t i = 0; i < m.mid.size(); ++i) {
        if (i % 2 == 0) keep = cur;                     // even hidden layer: remember its input for the next (odd) layer's residual
        int nxt = 0;
        while (nxt == cur || nxt == keep) ++nxt;
        GemmEpi g{};
        g.act = act; g.C = h[nxt]; g.ldc = ldh; g.rows_valid = rows_valid;
        const bool last = i + 1 == m.mid.size();
        if (all) {
            g.A16 = reinterpret_cast<const unsigned short*>(h[cur]);
            if (i % 2 == 1) { g.residual16 = reinterpret_cast<const unsigned short*>(h[keep]); g.ldr16 = m.mid[i].N_pad; }
            g.C = nullptr;
            g.C16 = last ? last_limbs : reinterpret_cast<unsigned short*>(h[nxt]);
            if (last) g.c16_scale = last_scale;
            if (wide_ok) {
                g.a16_scale = kOneAccActScale;
                if (g.residual16) g.r16_scale = kOneAccActScale;
                if (!last) g.c16_scale = kOneAccActScale;
            }
        } else {
            if (i % 2 == 1) { g.residual = h[keep]; g.ldr = ldh; }
            if (last_limbs && last) { g.C = nullptr; g.C16 = last_limbs; g.c16_scale = last_scale; }
        }
        ASeg a{h[cur], ldh};
        launch_gemm(m.mid[i], &a, rows, g, EPI_LINEAR, s);
        cur = nxt;
    }
    return last_limbs ? -1 : cur;
}
void attach_mlp_rows_images(DeviceArena& arena, PackedMLP& m) {
    auto fits = [](const PackedLinear& L) { return L.W2 && L.N_pad == 512 && L.K_pad <= 512 && L.K_pad % 16 == 0 && L.n_alloc >= 512 && L.bias; };
    if (m.mid.empty() || !fits(m.in_layer)) return;
    for (const PackedLinear& L : m.mid) if (!fits(L) || L.K_pad != 512 || L.nseg != 1) return;
    auto attach = [&](PackedLinear& L) {
        L.Wf = (unsigned short*)arena.alloc_bytes(mlp_rows_image_bytes(L.K_pad));
        launch_mlp_rows_image(L, L.Wf, nullptr);
    };
    attach(m.in_layer);
    for (PackedLinear& L : m.mid) attach(L);
    // round 4: the hidden layers' one-accumulator weight images for the 256 x 256 Linear kernel (spline_wide.hip EPI 1), natural row order
    for (PackedLinear& L : m.mid) spline_wide_attach(arena, L, L.wmax, nullptr, false);
}
int max_hidden_pad(const PackedMLP& m) {
    int mx = 0;
    for (int s : m.sizes) mx = std::max(mx, round_up(s, 32));
    return mx;
}

}  // namespace fc
