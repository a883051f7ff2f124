/* mgn_ref.c -- fp32 C restatement of MGN-spec v1.  TEST INFRASTRUCTURE + CPU BASELINE, NOT PRODUCT CODE.
 *
 * PARITY UNPINNED (see oracle/mgn_oracle.py header): the reference's arithmetic lives in GraphNetCore.jl
 * 0.3 / Lux 0.5, which are not under /root/reference and cannot be run here; this file restates the same
 * published algorithm (DeepMind MeshGraphNets, the model GraphNetCore implements per reference
 * README.md:9-19) and is pinned against oracle/mgn_oracle.py (float64) and the committed golden vectors.
 *
 * It computes exactly what the reference's CPU path computes for `mgn.model(graph, ps, st)` (reference
 * src/solve.jl:200; device = cpu_device(), src/MeshGraphNets.jl:260-262): Encoder -> mps x Processor ->
 * Decoder in Float32, written the plain way (explicit concat [v_s; v_r; e], one K=3L GEMM, scatter-add by
 * receiver) -- deliberately NOT the factored form the HIP kernels use, so it is an independent check.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parameter packing: MGN-spec order (include/mgn_hip.h): enc-node, enc-edge, (edge,node) x mps, decoder;
 * per MLP W1,b1,W2,b2,W3,b3,[gamma,beta]; W row-major [in][out].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LN_EPS 1e-5f
#define TB 64 /* rows per tile */

typedef struct {
    const float *W[3], *b[3], *gamma, *beta;
    int in, L, out, ln;
} mlp_t;

static const float* mlp_bind(mlp_t* m, const float* p, int in, int L, int out, int ln) {
    const int dims[4] = {in, L, L, out};
    m->in = in; m->L = L; m->out = out; m->ln = ln;
    for (int i = 0; i < 3; ++i) {
        m->W[i] = p; p += (size_t)dims[i] * dims[i + 1];
        m->b[i] = p; p += dims[i + 1];
    }
    m->gamma = m->beta = 0;
    if (ln) { m->gamma = p; p += out; m->beta = p; p += out; }
    return p;
}

/* Y[rows][n] = act(X[rows][k] * W[k][n] + b).  Register-blocked: RB rows x CB columns of Y are kept in vector
 * registers while k runs, so every W row segment loaded feeds RB FMAs (the un-blocked i-k-j loop is bound by
 * reloading Y).  Plain C, vectorised by gcc -O3 (AVX2/FMA at -march=x86-64-v3). */
static void dense(const float* X, int rows, int k, const float* W, const float* b, int n, int relu, float* Y) {
    enum { RB = 3, CB = 32 };   /* 12 ymm accumulators: the sweet spot of gcc's AVX2 code here (2.3x the un-blocked loop) */
    int i = 0;
    for (; i + RB <= rows; i += RB) {
        for (int j0 = 0; j0 < n; j0 += CB) {
            const int cb = (n - j0) < CB ? (n - j0) : CB;
            float acc[RB][CB];
            for (int r = 0; r < RB; ++r)
                for (int j = 0; j < cb; ++j) acc[r][j] = b[j0 + j];
            if (cb == CB) {
                for (int kk = 0; kk < k; ++kk) {
                    const float* w = W + (size_t)kk * n + j0;
                    float xr[RB];
                    for (int r = 0; r < RB; ++r) xr[r] = X[(size_t)(i + r) * k + kk];
                    for (int r = 0; r < RB; ++r)
                        for (int j = 0; j < CB; ++j) acc[r][j] += xr[r] * w[j];
                }
            } else {
                for (int kk = 0; kk < k; ++kk) {
                    const float* w = W + (size_t)kk * n + j0;
                    for (int r = 0; r < RB; ++r) {
                        const float xv = X[(size_t)(i + r) * k + kk];
                        for (int j = 0; j < cb; ++j) acc[r][j] += xv * w[j];
                    }
                }
            }
            for (int r = 0; r < RB; ++r) {
                float* y = Y + (size_t)(i + r) * n + j0;
                for (int j = 0; j < cb; ++j) y[j] = (relu && acc[r][j] < 0.f) ? 0.f : acc[r][j];
            }
        }
    }
    for (; i < rows; ++i) {   /* remainder rows */
        float* y = Y + (size_t)i * n;
        for (int j = 0; j < n; ++j) y[j] = b[j];
        const float* x = X + (size_t)i * k;
        for (int kk = 0; kk < k; ++kk) {
            const float xv = x[kk];
            const float* w = W + (size_t)kk * n;
            for (int j = 0; j < n; ++j) y[j] += xv * w[j];
        }
        if (relu) for (int j = 0; j < n; ++j) y[j] = y[j] > 0.f ? y[j] : 0.f;
    }
}

static void layer_norm(float* Y, int rows, int n, const float* gamma, const float* beta) {
    for (int i = 0; i < rows; ++i) {
        float* y = Y + (size_t)i * n;
        float mu = 0.f;
        for (int j = 0; j < n; ++j) mu += y[j];
        mu /= (float)n;
        float var = 0.f;
        for (int j = 0; j < n; ++j) { const float d = y[j] - mu; var += d * d; }
        var /= (float)n;
        const float rs = 1.0f / sqrtf(var + LN_EPS);
        for (int j = 0; j < n; ++j) y[j] = (y[j] - mu) * rs * gamma[j] + beta[j];
    }
}

/* out[rows][m.out] = MLP(X[rows][m.in]); scratch h1,h2 hold rows*L floats each */
static void mlp_apply(const mlp_t* m, const float* X, int rows, float* h1, float* h2, float* out) {
    dense(X, rows, m->in, m->W[0], m->b[0], m->L, 1, h1);
    dense(h1, rows, m->L, m->W[1], m->b[1], m->L, 1, h2);
    dense(h2, rows, m->L, m->W[2], m->b[2], m->out, 0, out);
    if (m->ln) layer_norm(out, rows, m->out, m->gamma, m->beta);
}

typedef struct {
    int Fn, Fe, O, L, mps;
    mlp_t enc_node, enc_edge, dec, *pe, *pn;
} model_t;

static int model_bind(model_t* M, const float* p, int Fn, int Fe, int O, int L, int mps) {
    M->Fn = Fn; M->Fe = Fe; M->O = O; M->L = L; M->mps = mps;
    M->pe = (mlp_t*)malloc(sizeof(mlp_t) * (size_t)mps);
    M->pn = (mlp_t*)malloc(sizeof(mlp_t) * (size_t)mps);
    if (!M->pe || !M->pn) return -1;
    p = mlp_bind(&M->enc_node, p, Fn, L, L, 1);
    p = mlp_bind(&M->enc_edge, p, Fe, L, L, 1);
    for (int k = 0; k < mps; ++k) {
        p = mlp_bind(&M->pe[k], p, 3 * L, L, L, 1);
        p = mlp_bind(&M->pn[k], p, 2 * L, L, L, 1);
    }
    mlp_bind(&M->dec, p, L, L, O, 0);
    return 0;
}

static void model_free(model_t* M) { free(M->pe); free(M->pn); }

/* CSR by receiver: for node n, edges idx[ptr[n]..ptr[n+1]) in input order */
static int build_csr(int N, int64_t E, const int32_t* rcv, int64_t** ptr_out, int64_t** idx_out) {
    int64_t* ptr = (int64_t*)calloc((size_t)N + 1, sizeof(int64_t));
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)(E > 0 ? E : 1));
    int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    if (!ptr || !idx || !cur) return -1;
    for (int64_t i = 0; i < E; ++i) ptr[rcv[i] + 1]++;
    for (int n = 0; n < N; ++n) ptr[n + 1] += ptr[n];
    for (int n = 0; n < N; ++n) cur[n] = ptr[n];
    for (int64_t i = 0; i < E; ++i) idx[cur[rcv[i]]++] = i;
    free(cur);
    *ptr_out = ptr; *idx_out = idx;
    return 0;
}

/* one processor step (DeepMind GraphNetBlock order): e' = MLP_e([v_s; v_r; e]); agg = scatter_add(e');
 * v' = MLP_v([v; agg]); v += v'; e += e'.  enew is an [E][L] scratch that receives e'. */
static void processor_step(const model_t* M, int k, int N, int64_t E, const int32_t* snd, const int32_t* rcv,
                           const int64_t* ptr, const int64_t* idx, float* v, float* e, float* enew) {
    const int L = M->L;
    const int64_t etiles = (E + TB - 1) / TB;
#pragma omp parallel
    {
        float* X = (float*)malloc(sizeof(float) * TB * 3 * (size_t)L);
        float* h1 = (float*)malloc(sizeof(float) * TB * (size_t)L);
        float* h2 = (float*)malloc(sizeof(float) * TB * (size_t)L);
#pragma omp for schedule(static)
        for (int64_t t = 0; t < etiles; ++t) {
            const int64_t e0 = t * TB;
            const int rows = (int)((E - e0) < TB ? (E - e0) : TB);
            for (int i = 0; i < rows; ++i) {
                memcpy(X + (size_t)i * 3 * L, v + (size_t)snd[e0 + i] * L, sizeof(float) * (size_t)L);
                memcpy(X + (size_t)i * 3 * L + L, v + (size_t)rcv[e0 + i] * L, sizeof(float) * (size_t)L);
                memcpy(X + (size_t)i * 3 * L + 2 * L, e + (size_t)(e0 + i) * L, sizeof(float) * (size_t)L);
            }
            mlp_apply(&M->pe[k], X, rows, h1, h2, enew + (size_t)e0 * L);
        }
        const int ntiles = (N + TB - 1) / TB;
#pragma omp for schedule(static)
        for (int t = 0; t < ntiles; ++t) {
            const int n0 = t * TB;
            const int rows = (N - n0) < TB ? (N - n0) : TB;
            float* Y = X + (size_t)TB * 2 * L; /* reuse tail of X as output scratch */
            for (int i = 0; i < rows; ++i) {
                float* x = X + (size_t)i * 2 * L;
                memcpy(x, v + (size_t)(n0 + i) * L, sizeof(float) * (size_t)L);
                float* a = x + L;
                for (int j = 0; j < L; ++j) a[j] = 0.f;
                for (int64_t q = ptr[n0 + i]; q < ptr[n0 + i + 1]; ++q) {
                    const float* r = enew + (size_t)idx[q] * L;
                    for (int j = 0; j < L; ++j) a[j] += r[j];
                }
            }
            mlp_apply(&M->pn[k], X, rows, h1, h2, Y);
            /* all gathers of v for this step finished in the edge pass (implicit barrier above), and the
             * node pass reads only its own rows, so v can be updated in place */
            for (int i = 0; i < rows; ++i) {
                float* vr = v + (size_t)(n0 + i) * L;
                for (int j = 0; j < L; ++j) vr[j] += Y[(size_t)i * L + j];
            }
        }
#pragma omp for schedule(static)
        for (int64_t i = 0; i < E * (int64_t)L; ++i) e[i] += enew[i];
        free(X); free(h1); free(h2);
    }
}

int mgn_ref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* nsteps processor steps on latents v [N][L], e [E][L] (in place).  0-based indices. */
int mgn_ref_processor_steps(const float* params, int Fn, int Fe, int O, int L, int mps, int N, int64_t E,
                            const int32_t* snd, const int32_t* rcv, float* v, float* e, int nsteps) {
    model_t M;
    if (nsteps > mps || model_bind(&M, params, Fn, Fe, O, L, mps)) return -1;
    int64_t *ptr = 0, *idx = 0;
    if (build_csr(N, E, rcv, &ptr, &idx)) return -1;
    float* enew = (float*)malloc(sizeof(float) * (size_t)(E > 0 ? E : 1) * (size_t)L);
    if (!enew) return -1;
    for (int k = 0; k < nsteps; ++k) processor_step(&M, k, N, E, snd, rcv, ptr, idx, v, e, enew);
    free(enew); free(ptr); free(idx);
    model_free(&M);
    return 0;
}

/* full forward: nf [N][Fn], ef [E][Fe] -> out [N][O] */
int mgn_ref_forward(const float* params, int Fn, int Fe, int O, int L, int mps, int N, int64_t E, const int32_t* snd,
                    const int32_t* rcv, const float* nf, const float* ef, float* out) {
    model_t M;
    if (model_bind(&M, params, Fn, Fe, O, L, mps)) return -1;
    int64_t *ptr = 0, *idx = 0;
    if (build_csr(N, E, rcv, &ptr, &idx)) return -1;
    float* v = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * (size_t)L);
    float* e = (float*)malloc(sizeof(float) * (size_t)(E > 0 ? E : 1) * (size_t)L);
    float* enew = (float*)malloc(sizeof(float) * (size_t)(E > 0 ? E : 1) * (size_t)L);
    if (!v || !e || !enew) return -1;
#pragma omp parallel
    {
        float* h1 = (float*)malloc(sizeof(float) * TB * (size_t)L);
        float* h2 = (float*)malloc(sizeof(float) * TB * (size_t)L);
#pragma omp for schedule(static)
        for (int t = 0; t < (N + TB - 1) / TB; ++t) {
            const int n0 = t * TB, rows = (N - n0) < TB ? (N - n0) : TB;
            mlp_apply(&M.enc_node, nf + (size_t)n0 * Fn, rows, h1, h2, v + (size_t)n0 * L);
        }
#pragma omp for schedule(static)
        for (int64_t t = 0; t < (E + TB - 1) / TB; ++t) {
            const int64_t e0 = t * TB;
            const int rows = (int)((E - e0) < TB ? (E - e0) : TB);
            mlp_apply(&M.enc_edge, ef + (size_t)e0 * Fe, rows, h1, h2, e + (size_t)e0 * L);
        }
        free(h1); free(h2);
    }
    for (int k = 0; k < mps; ++k) processor_step(&M, k, N, E, snd, rcv, ptr, idx, v, e, enew);
#pragma omp parallel
    {
        float* h1 = (float*)malloc(sizeof(float) * TB * (size_t)L);
        float* h2 = (float*)malloc(sizeof(float) * TB * (size_t)L);
#pragma omp for schedule(static)
        for (int t = 0; t < (N + TB - 1) / TB; ++t) {
            const int n0 = t * TB, rows = (N - n0) < TB ? (N - n0) : TB;
            mlp_apply(&M.dec, v + (size_t)n0 * L, rows, h1, h2, out + (size_t)n0 * O);
        }
        free(h1); free(h2);
    }
    free(v); free(e); free(enew); free(ptr); free(idx);
    model_free(&M);
    return 0;
}
