// Host-side optimiser (glmmrmcml_amd/csrc/optim.hip: BOBYQA family, R-style finite differences) compiled as
// plain C++ with AddressSanitizer + UndefinedBehaviorSanitizer and driven over unbounded, bounded and
// start-on-bound problems.  Built and run by tests/test_host_sanitizers.py (CPU only: GPU sanitizers are not
// available on this pool).
#include "optim.h"
#include <cstdio>
#include <cmath>
using namespace mcml;
int main() {
    int fails = 0;
    for (int n : {2, 3, 5, 8, 13}) {
        objective_fn rosen = [n](const std::vector<double>& x, double* v) {
            double s = 0; for (int i = 0; i + 1 < n; ++i) s += 100 * pow(x[i + 1] - x[i] * x[i], 2) + pow(1 - x[i], 2);
            *v = s; return 0; };
        std::vector<double> x0(n, -1.2), lo(n, -HUGE_VAL), up(n, HUGE_VAL);
        BobyqaOpts o; o.maxfun = 20000;
        BobyqaResult r;
        int rc = bobyqa(rosen, x0, lo, up, o, &r);
        printf("rosen n=%d rc=%d f=%.3e nfev=%d\n", n, rc, r.fval, r.nfev);
        // (the 5-dimensional chained Rosenbrock has a genuine local minimum f = 3.93 reachable from -1.2)
        if (rc || (r.fval > 1e-6 && !(n == 5 && fabs(r.fval - 3.9308) < 1e-3))) ++fails;
        // bounded: minimum on the boundary
        std::vector<double> lo2(n, 1.5), up2(n, 4.0), x1(n, 2.0);
        rc = bobyqa(rosen, x1, lo2, up2, o, &r);
        printf("  bounded rc=%d f=%.6f nfev=%d x0=%.4f\n", rc, r.fval, r.nfev, r.x[0]);
        if (rc) ++fails;
        // start on / near bounds
        std::vector<double> lo3(n, 1e-6), up3(n, HUGE_VAL), x2(n, 1e-6 + 1e-18);
        objective_fn quad = [n](const std::vector<double>& x, double* v) { double s = 0; for (int i = 0; i < n; ++i) s += (x[i] - 0.3 * (i + 1)) * (x[i] - 0.3 * (i + 1)); *v = s; return 0; };
        rc = bobyqa(quad, x2, lo3, up3, o, &r);
        printf("  from-bound rc=%d f=%.3e nfev=%d\n", rc, r.fval, r.nfev);
        if (rc || r.fval > 1e-8) ++fails;
        std::vector<double> H, nd(n, 1e-4), none;
        rc = fd_hessian(quad, std::vector<double>(n, 0.5), nd, false, none, none, &H);
        if (rc || fabs(H[0] - 2.0) > 1e-5) ++fails;
    }
    printf("fails=%d\n", fails);
    return fails;
}
