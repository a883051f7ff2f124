/* The boundary from plain C: include/mgn_hip.h must compile as C99 (-pedantic) and the library must be callable without any
 * C++ or Python in between.  Host-only handle (MGN_DEVICE_NONE): partitioner, graph prologue helpers, record reader;
 * every compute entry point has to refuse.  Built and run by tests/test_abi_and_host.py.                              */
#include <stdio.h>
#include <string.h>

#include "../../include/mgn_hip.h"

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(void) {
    mgn_config cfg;
    mgn_handle* h = NULL;
    /* two triangles sharing an edge: 5 undirected, 10 directed edges (KAT-1) */
    const int32_t cells[6] = {0, 1, 2, 1, 3, 2};
    int32_t snd[10], rcv[10], n_own = 0, n_halo = 0;
    int64_t n_dir = 0, e_local = 0;
    float nf[4 * 9], ef[10 * 3], out[4 * 2];

    memset(&cfg, 0, sizeof cfg);
    cfg.Fn = 9; cfg.Fe = 3; cfg.O = 2; cfg.L = 128; cfg.hidden_layers = 2; cfg.mps = 15;
    cfg.dtype = MGN_F32; cfg.rank = 0; cfg.nranks = 1; cfg.device = MGN_DEVICE_NONE;
    CHECK(mgn_param_count(&cfg) == 2332674u);                 /* 34 560 + 33 792 + 15 x (82 560 + 66 176) + 33 282 floats */
    CHECK(mgn_create(&cfg, &h) == MGN_OK && h != NULL);
    CHECK(mgn_triangles_to_edges(cells, 2, NULL, NULL, &n_dir) == MGN_OK && n_dir == 10);
    CHECK(mgn_triangles_to_edges(cells, 2, snd, rcv, &n_dir) == MGN_OK);
    CHECK(snd[0] == 1 && rcv[0] == 0 && snd[5] == 0 && rcv[5] == 1);   /* first-occurrence order, then the reversed copies */
    CHECK(mgn_set_graph(h, 4, 10, snd, rcv, 0, NULL, 0) == MGN_OK);
    CHECK(mgn_partition_info(h, &n_own, &n_halo, &e_local) == MGN_OK && n_own == 4 && n_halo == 0 && e_local == 10);
    memset(nf, 0, sizeof nf); memset(ef, 0, sizeof ef);
    CHECK(mgn_forward(h, nf, ef, out) == MGN_E_HIP);          /* no CPU compute path */
    CHECK(strstr(mgn_last_error(h), "host-only") != NULL);
    CHECK(mgn_set_graph(h, 4, 10, snd, rcv, 7, NULL, 0) == MGN_E_ARG);
    CHECK(mgn_crc32c("123456789", 9) == 0xE3069283u);
    cfg.L = 100;
    {
        mgn_handle* bad = NULL;
        CHECK(mgn_create(&cfg, &bad) == MGN_E_ARG && bad == NULL);
        CHECK(strstr(mgn_last_error(NULL), "L must be") != NULL);
    }
    mgn_destroy(h);
    printf("abi_check OK\n");
    return 0;
}
