// dsim_layout_tool.cpp -- TEST / BUILD-TIME tool: exposes the layout builder of diffrl_amd/csrc/dsim_layout.hpp alone
// (no kernel code), so that tools/gen_static_layouts.py can regenerate dsim_static_layouts.hpp even while the phase
// code that depends on that generated header does not compile yet.
#include <cstring>

#include "../../diffrl_amd/csrc/dsim_layout.hpp"

extern "C" int dsim_emu_layout(const dsim_model_desc* m, int* off_out, int n_off, int* dims_out) {
    DsimLayout lay;
    std::string err = dsim_build_layout(*m, lay);
    if (!err.empty()) return -1;
    const int n = (int)(sizeof(DsimOff) / sizeof(int));
    if (n_off < n) return n;
    memcpy(off_out, &lay.o, sizeof(DsimOff));
    memcpy(dims_out, &lay.d, sizeof(DsimDims));
    return n;
}
