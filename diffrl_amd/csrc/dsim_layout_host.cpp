// dsim_layout_host.cpp -- HOST-ONLY build of the layout builder (dsim_layout.hpp, the one dsim_model_create runs): no kernels,
// no HIP.  `python -m diffrl_amd.specialise` calls it to compute the compile-time tables of a user model's specialised kernel
// set without a GPU (g++ -shared -fPIC dsim_layout_host.cpp -o libdsim_layout_host.so, one second).
#include <cstring>

#include "dsim_layout.hpp"

// out = [DsimOff as ints | DsimDims as ints] -- exactly what match_variant (dsim_hip.hip) compares a model's run-time layout
// with.  Returns the number of ints (call with cap = 0 to ask), or -1 and *err_out (static storage) if the model is refused.
extern "C" int dsim_layout_dump(const dsim_model_desc* desc, int* out, int cap, int* n_off, const char** err_out) {
    static thread_local std::string err;
    DsimLayout lay;
    err = dsim_build_layout(*desc, lay);
    if (!err.empty()) {
        if (err_out) *err_out = err.c_str();
        return -1;
    }
    const int no = (int)(sizeof(DsimOff) / sizeof(int)), nd = (int)(sizeof(DsimDims) / sizeof(int));
    if (n_off) *n_off = no;
    if (cap >= no + nd) {
        memcpy(out, &lay.o, sizeof(DsimOff));
        memcpy(out + no, &lay.d, sizeof(DsimDims));
    }
    return no + nd;
}
