// Model constants: upload + derived tables (CSC regressor, per-bone vertex lists, joint depths).
#include <algorithm>
#include <cstring>
#include <numeric>
#include <string>

#include "common.h"

static thread_local char g_err[512] = "";

void smil_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *smil_last_error(void) { return g_err; }
// "instrumented" marks a library built by `make variant` for tools/dbg (timers, counters, cut-off experiments - possibly with garbage
// results by design): tests/test_abi_cpu.py checks that the library the product loads is not one of those.
#ifdef SMIL_INSTRUMENTED
extern "C" const char *smil_version(void) { return "smilfit 0.3 (gfx950) instrumented"; }
#else
extern "C" const char *smil_version(void) { return "smilfit 0.3 (gfx950)"; }
#endif

template <typename T>
static int upload(SmilModel *m, T **dst, const T *src, size_t n) {
    *dst = nullptr;
    if (n == 0) n = 1;  // keep pointers non-null
    void *p = nullptr;
    SMIL_HIP(hipMalloc(&p, n * sizeof(T)));
    m->allocations.push_back(p);
    if (src) SMIL_HIP(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    else SMIL_HIP(hipMemset(p, 0, n * sizeof(T)));
    *dst = static_cast<T *>(p);
    return SMIL_OK;
}

extern "C" int smil_model_create(const SmilModelDesc *d, SmilModel **out) {
    SMIL_REQUIRE(d && out, "smil_model_create: null argument");
    SMIL_REQUIRE(d->V > 0 && d->F > 0 && d->J > 0 && d->nB >= 0, "smil_model_create: bad sizes V=%d F=%d J=%d nB=%d",
                 d->V, d->F, d->J, d->nB);
    SMIL_REQUIRE(d->J <= SMIL_MAX_JOINTS, "smil_model_create: J=%d exceeds %d", d->J, SMIL_MAX_JOINTS);
    SMIL_REQUIRE(d->nB <= SMIL_MAX_BETAS, "smil_model_create: nB=%d exceeds %d", d->nB, SMIL_MAX_BETAS);
    SMIL_REQUIRE(d->v_template && d->faces && d->parents && d->skin_idx && d->skin_w && d->jreg_rowptr,
                 "smil_model_create: null table");
    SMIL_REQUIRE(d->nB == 0 || d->shapedirs, "smil_model_create: shapedirs missing");
    SMIL_REQUIRE(!d->static_joints || d->J_static, "smil_model_create: static joints without J table");
    const int V = d->V, F = d->F, J = d->J;
    SMIL_REQUIRE(d->parents[0] == -1, "smil_model_create: joint 0 must be the root");
    std::vector<int> depth(J, 0);
    int max_depth = 0;
    for (int i = 1; i < J; ++i) {
        SMIL_REQUIRE(d->parents[i] >= 0 && d->parents[i] < i, "smil_model_create: parent[%d]=%d does not precede it", i,
                     d->parents[i]);
        depth[i] = depth[d->parents[i]] + 1;
        max_depth = std::max(max_depth, depth[i]);
    }
    for (int i = 0; i < 3 * F; ++i)
        SMIL_REQUIRE(d->faces[i] >= 0 && d->faces[i] < V, "smil_model_create: face index %d out of range", d->faces[i]);
    int max_valence = 0;
    {
        std::vector<int> valence(V, 0);
        for (int i = 0; i < 3 * F; ++i) max_valence = std::max(max_valence, ++valence[d->faces[i]]);
    }
    const int nnz = d->jreg_rowptr[J];
    SMIL_REQUIRE(nnz >= 0 && (nnz == 0 || (d->jreg_col && d->jreg_val)), "smil_model_create: regressor CSR malformed");
    for (int e = 0; e < nnz; ++e)
        SMIL_REQUIRE(d->jreg_col[e] >= 0 && d->jreg_col[e] < V, "smil_model_create: regressor column out of range");

    // packed skin table + per-bone lists
    std::vector<uint32_t> packed(V);
    std::vector<float4> w4(V);
    std::vector<int> bone_cnt(J + 1, 0);
    for (int v = 0; v < V; ++v) {
        uint32_t p = 0;
        float w[4];
        for (int k = 0; k < 4; ++k) {
            const int id = d->skin_idx[4 * v + k];
            SMIL_REQUIRE(id >= 0 && id < J, "smil_model_create: bone id %d out of range at vertex %d", id, v);
            p |= (uint32_t)id << (8 * k);
            w[k] = d->skin_w[4 * v + k];
            if (w[k] != 0.f) bone_cnt[id + 1]++;
        }
        packed[v] = p;
        w4[v] = make_float4(w[0], w[1], w[2], w[3]);
    }
    std::vector<int> bone_ptr(J + 1, 0);
    std::partial_sum(bone_cnt.begin(), bone_cnt.end(), bone_ptr.begin());
    const int bone_nnz = bone_ptr[J];
    std::vector<int> bone_vid(std::max(bone_nnz, 1)), fill(bone_ptr.begin(), bone_ptr.end() - 1);
    std::vector<float> bone_w(std::max(bone_nnz, 1));
    for (int v = 0; v < V; ++v)
        for (int k = 0; k < 4; ++k) {
            const float w = d->skin_w[4 * v + k];
            if (w == 0.f) continue;
            const int id = d->skin_idx[4 * v + k];
            bone_vid[fill[id]] = v;
            bone_w[fill[id]] = w;
            fill[id]++;
        }
    // CSC of the joint regressor
    std::vector<int> colptr(V + 1, 0), crow(std::max(nnz, 1));
    std::vector<float> cval(std::max(nnz, 1));
    for (int e = 0; e < nnz; ++e) colptr[d->jreg_col[e] + 1]++;
    std::partial_sum(colptr.begin(), colptr.end(), colptr.begin());
    {
        std::vector<int> pos(colptr.begin(), colptr.end() - 1);
        for (int j = 0; j < J; ++j)
            for (int e = d->jreg_rowptr[j]; e < d->jreg_rowptr[j + 1]; ++e) {
                const int v = d->jreg_col[e];
                crow[pos[v]] = j;
                cval[pos[v]] = d->jreg_val[e];
                pos[v]++;
            }
    }

    // J_regressor @ shapedirs[k] (nB,J,3) and the bones by falling list length
    std::vector<float> jreg_shape((size_t)std::max(d->nB, 1) * J * 3, 0.f);
    for (int k = 0; k < d->nB; ++k)
        for (int j = 0; j < J; ++j)
            for (int e = d->jreg_rowptr[j]; e < d->jreg_rowptr[j + 1]; ++e)
                for (int c = 0; c < 3; ++c)
                    jreg_shape[((size_t)k * J + j) * 3 + c] += d->jreg_val[e] * d->shapedirs[(size_t)k * 3 * V + 3 * d->jreg_col[e] + c];
    std::vector<int2> vfirst(V);
    for (int v = 0; v < V; ++v) {
        const int n_ent = colptr[v + 1] - colptr[v];
        SMIL_REQUIRE(n_ent < 32768, "smil_model_create: vertex %d has %d regressor entries", v, n_ent);
        float w0 = n_ent ? cval[colptr[v]] : 0.f;
        int wb;
        std::memcpy(&wb, &w0, 4);
        vfirst[v] = make_int2((n_ent ? crow[colptr[v]] : 0) | (n_ent << 16), wb);
    }
    // the bone lists dealt to BONE_WAVES waves: longest first, each to the wave with the least work so far (work = 64-entry
    // segments + one for the reduction at the end of a bone)
    std::vector<int> by_len(J);
    std::iota(by_len.begin(), by_len.end(), 0);
    std::stable_sort(by_len.begin(), by_len.end(), [&](int a, int b) { return bone_cnt[a + 1] > bone_cnt[b + 1]; });
    std::vector<std::vector<int>> per_wave(BONE_WAVES);
    {
        std::vector<int> load(BONE_WAVES, 0);
        for (int j : by_len) {
            const int w = (int)(std::min_element(load.begin(), load.end()) - load.begin());
            per_wave[w].push_back(j);
            load[w] += (bone_cnt[j + 1] + WAVE - 1) / WAVE + 1;
        }
    }
    size_t deepest = 0;
    for (const auto &l : per_wave) deepest = std::max(deepest, l.size());
    const int bone_slots = BONE_WAVES * (int)deepest;
    std::vector<int> bone_order(std::max(bone_slots, 1), -1);
    for (int w = 0; w < BONE_WAVES; ++w)
        for (size_t k = 0; k < per_wave[w].size(); ++k) bone_order[w + k * BONE_WAVES] = per_wave[w][k];

    SmilModel *m = new SmilModel();
    m->V = V; m->F = F; m->J = J; m->nB = d->nB;
    m->max_depth = max_depth;
    m->max_valence = max_valence;
    m->static_joints = d->static_joints != 0;
    m->jreg_nnz = nnz;
    m->bone_nnz = bone_nnz;
    int rc = SMIL_OK;
#define UP(field, src, n) if ((rc = upload(m, &m->field, src, (size_t)(n))) != SMIL_OK) { smil_model_destroy(m); return rc; }
    UP(v_template, d->v_template, 3 * V);
    UP(shapedirs, d->shapedirs, (size_t)d->nB * 3 * V);
    UP(faces, d->faces, 3 * F);
    UP(parents, d->parents, J);
    UP(depth, depth.data(), J);
    UP(skin_idx, packed.data(), V);
    UP(skin_w, w4.data(), V);
    UP(jreg_rowptr, d->jreg_rowptr, J + 1);
    UP(jreg_col, d->jreg_col, nnz);
    UP(jreg_val, d->jreg_val, nnz);
    UP(jreg_colptr, colptr.data(), V + 1);
    UP(jreg_row, crow.data(), nnz);
    UP(jreg_cval, cval.data(), nnz);
    UP(bone_ptr, bone_ptr.data(), J + 1);
    UP(bone_vid, bone_vid.data(), bone_nnz);
    UP(bone_w, bone_w.data(), bone_nnz);
    UP(jreg_vfirst, vfirst.data(), V);
    UP(jreg_shape, jreg_shape.data(), (size_t)std::max(d->nB, 1) * J * 3);
    UP(bone_order, bone_order.data(), bone_slots);
    m->bone_slots = bone_slots;
    UP(J_static, d->static_joints ? d->J_static : (const float *)nullptr, 3 * J);
    if (d->posedirs) { UP(posedirs, d->posedirs, (size_t)9 * (J - 1) * 3 * V); }
#undef UP
    *out = m;
    return SMIL_OK;
}

extern "C" void smil_model_destroy(SmilModel *m) {
    if (!m) return;
    for (void *p : m->allocations) (void)hipFree(p);
    delete m;
}

extern "C" int smil_model_dims(const SmilModel *m, int32_t dims[4]) {
    SMIL_REQUIRE(m && dims, "smil_model_dims: null argument");
    dims[0] = m->V; dims[1] = m->F; dims[2] = m->J; dims[3] = m->nB;
    return SMIL_OK;
}
