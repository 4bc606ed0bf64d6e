/* The C ABI used from plain C (C99, no C++ anywhere on the caller's side): what a binding in any
 * language with a C FFI does.  Solves -lap(phi) = rho/eps0 on n^3 nodes of the reference driver's
 * box (cpp/Main.cpp:37) with 0 V walls and the synthetic charge density of this repository's
 * benchmarks, then differentiates phi on the device.
 *
 *   c_abi_example [n=41] [depth=2] [tol=1e-6] [phi.bin]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../include/mg_hip.h"

#define EPS_0 8.85418782e-12 /* cpp/World.h:14 */

static int check(int rc, const char *what)
{
    if (rc != MG_OK) fprintf(stderr, "c_abi_example: %s: %s\n", what, mg_last_error());
    return rc;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 41, depth = argc > 2 ? atoi(argv[2]) : 2;
    const double tol = argc > 3 ? atof(argv[3]) : 1e-6;
    const double dh[3] = {0.4 / (n - 1), 0.4 / (n - 1), 0.4 / (n - 1)};
    const double pi = acos(-1.0);
    const size_t nodes = (size_t)n * n * n;
    double *rho, *phi, *ef, l2 = 0;
    mg_config cfg;
    mg_handle *h = NULL;
    uint64_t bytes = 0;
    int converged = 0, i, j, k, rc = 1;
    unsigned cycles = 0;

    if (n < 3 || mg_abi_version() != MG_ABI_VERSION) return 2;
    if (check(mg_default_config(&cfg, n, n, n, dh, depth), "mg_default_config")) return 1;
    cfg.flags |= MG_FLAG_RESERVE_EF;
    if (check(mg_device_bytes(&cfg, &bytes), "mg_device_bytes")) return 1;
    printf("%d^3, %d coarse grids: %.1f MB of device memory\n", n, depth, bytes / 1e6);
    rho = (double *)malloc(nodes * sizeof(double));
    phi = (double *)calloc(nodes, sizeof(double));
    ef = (double *)malloc(nodes * 3 * sizeof(double));
    if (!rho || !phi || !ef) return 1;
    for (i = 0; i < n; i++)
        for (j = 0; j < n; j++)
            for (k = 0; k < n; k++) {
                const double x = (double)i / (n - 1), y = (double)j / (n - 1), z = (double)k / (n - 1);
                rho[((size_t)i * n + j) * n + k] = EPS_0 * 1000.0 * (sin(pi * x) * sin(pi * y) * sin(pi * z) +
                                                                      0.5 * sin(5 * pi * x) * sin(3 * pi * y) * sin(7 * pi * z));
            }
    if (check(mg_create(&cfg, &h), "mg_create")) goto done;
    if (check(mg_set_rho(h, rho), "mg_set_rho") || check(mg_set_phi(h, phi), "mg_set_phi")) goto done;
    if (check(mg_solve(h, 10000, tol, &converged, &cycles, &l2), "mg_solve")) goto done;
    if (check(mg_get_phi(h, phi), "mg_get_phi")) goto done;
    if (check(mg_compute_ef(h), "mg_compute_ef") || check(mg_get_ef(h, ef), "mg_get_ef")) goto done;
    printf("converged=%d cycles=%u L2=%.6e phi_centre=%.9e Ex_centre=%.9e\n", converged, cycles, l2,
           phi[((size_t)(n / 2) * n + n / 2) * n + n / 2], ef[(((size_t)(n / 2) * n + n / 2) * n + n / 2) * 3]);
    if (argc > 4) {
        FILE *f = fopen(argv[4], "wb");
        if (!f || fwrite(phi, sizeof(double), nodes, f) != nodes) {
            fprintf(stderr, "c_abi_example: cannot write %s\n", argv[4]);
            goto done;
        }
        fclose(f);
    }
    rc = converged ? 0 : 3;
done:
    mg_destroy(h);
    free(rho);
    free(phi);
    free(ef);
    return rc;
}
