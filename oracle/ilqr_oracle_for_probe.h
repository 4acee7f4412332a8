// TEST INFRASTRUCTURE: declaration of the oracle entry used by tools/probes (defined in oracle_probe_glue.cpp)
#pragma once
void oracle_linearize(const double* x, const double* u, double h, const double* g, double* A, double* B);
