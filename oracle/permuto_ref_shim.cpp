// extern "C" wrappers around the reference's vendored Permutohedral class (TEST INFRASTRUCTURE).
// Compiled together with /root/reference/third_party/permutohedral/permutohedral.cpp into
// oracle/_ref/libpermuto_ref.so by oracle/Makefile; mirrors the pybind surface of
// probreg/cc/permutohedral_lattice_py.cc:13-21 (init / get_lattice_size / filter).
#include "permutohedral.h"

extern "C" {

void* permuto_ref_create() { return new Permutohedral(); }
void permuto_ref_destroy(void* p) { delete static_cast<Permutohedral*>(p); }

// features: d x n column-major (== n x d row-major points)
void permuto_ref_init(void* p, const float* features, int d, int n, int with_blur) {
    MatrixXf f(d, n);
    for (size_t i = 0; i < (size_t)d * n; ++i) f.data()[i] = features[i];
    static_cast<Permutohedral*>(p)->init(f, with_blur != 0);
}

int permuto_ref_lattice_size(void* p) { return static_cast<Permutohedral*>(p)->getLatticeSize(); }

// values: ch x n column-major (== n x ch row-major); out same shape.  start is dropped by the reference
// (permutohedral.cpp:608-616) and therefore not even passed here.
void permuto_ref_filter(void* p, const float* values, int ch, int n, float* out) {
    MatrixXf v(ch, n), o(ch, n);
    for (size_t i = 0; i < (size_t)ch * n; ++i) v.data()[i] = values[i];
    static_cast<Permutohedral*>(p)->compute(o, v, false, 0);
    for (size_t i = 0; i < (size_t)ch * n; ++i) out[i] = o.data()[i];
}
}
