/* The compute entry points from plain C (C99, no C++ / Python between the program and libmgn_hip.so): create a handle on device 0,
 * install parameters and a graph, run mgn_forward (reference: mgn.model(graph, ps, st), src/solve.jl:200) and mgn_step (step!,
 * src/strategies.jl:418-422), write the results for tests/test_gpu_parity.py::test_compute_from_plain_c to compare with the oracle
 * and with the same calls made through ctypes.
 *   in.bin : int32 N, E, P, mps, nmask | int32 snd[E], rcv[E], mask[nmask] | float params[P], nf[N][9], ef[E][3], target[N][2]
 *   out.bin: float out[N][2], loss, grads[P]                                                                                  */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mgn_hip.h"

#define CHECK(cond)                                                                                   \
    do {                                                                                              \
        if (!(cond)) {                                                                                \
            fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, mgn_last_error(h)); \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

static int read_all(FILE* f, void* p, size_t bytes) { return fread(p, 1, bytes, f) == bytes; }

int main(int argc, char** argv) {
    mgn_config cfg;
    mgn_handle* h = NULL;
    int32_t hdr[5];
    int32_t *snd, *rcv, *mask;
    float *params, *nf, *ef, *target, *out, *grads, loss = 0.f;
    size_t N, E, P, nmask;
    FILE* f;
    if (argc != 3) return 2;
    f = fopen(argv[1], "rb");
    if (!f || !read_all(f, hdr, sizeof hdr)) return 2;
    N = (size_t)hdr[0]; E = (size_t)hdr[1]; P = (size_t)hdr[2]; nmask = (size_t)hdr[4];
    snd = malloc(E * 4); rcv = malloc(E * 4); mask = malloc(nmask * 4);
    params = malloc(P * 4); nf = malloc(N * 9 * 4); ef = malloc(E * 3 * 4); target = malloc(N * 2 * 4);
    out = malloc(N * 2 * 4); grads = malloc(P * 4);
    if (!snd || !rcv || !mask || !params || !nf || !ef || !target || !out || !grads) return 2;
    if (!read_all(f, snd, E * 4) || !read_all(f, rcv, E * 4) || !read_all(f, mask, nmask * 4) || !read_all(f, params, P * 4) ||
        !read_all(f, nf, N * 9 * 4) || !read_all(f, ef, E * 3 * 4) || !read_all(f, target, N * 2 * 4))
        return 2;
    fclose(f);

    memset(&cfg, 0, sizeof cfg);
    cfg.Fn = 9; cfg.Fe = 3; cfg.O = 2; cfg.L = 128; cfg.hidden_layers = 2; cfg.mps = hdr[3];
    cfg.dtype = MGN_F32; cfg.rank = 0; cfg.nranks = 1; cfg.device = 0;
    CHECK(mgn_param_count(&cfg) == P);
    CHECK(mgn_create(&cfg, &h) == MGN_OK && h != NULL);
    CHECK(mgn_set_params(h, params, P) == MGN_OK);
    CHECK(mgn_set_graph(h, (int32_t)N, (int64_t)E, snd, rcv, 0, NULL, 0) == MGN_OK);
    CHECK(mgn_forward(h, nf, ef, out) == MGN_OK);
    CHECK(mgn_step(h, nf, ef, target, mask, (int64_t)nmask, 0, grads, P, &loss) == MGN_OK);
    CHECK(mgn_forward(h, nf, NULL, out) == MGN_E_ARG);              /* errors come back as codes + text, not as crashes */
    CHECK(strlen(mgn_last_error(h)) > 0);
    mgn_destroy(h);
    h = NULL;

    f = fopen(argv[2], "wb");
    if (!f) return 2;
    fwrite(out, 4, N * 2, f);
    fwrite(&loss, 4, 1, f);
    fwrite(grads, 4, P, f);
    fclose(f);
    printf("abi_gpu OK\n");
    return 0;
}
