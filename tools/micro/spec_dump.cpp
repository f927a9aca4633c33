// Diagnostic: print the specialised kernel source the library would hand to hipRTC for a table read
// from stdin ("C", then per channel "K idx... w...").   usage: spec_dump [rr] [dd] [la] < table.txt
#include "../../vndecorrelate_amd/csrc/vnd_spec.hpp"
#include <iostream>
int main(int argc, char **argv)
{
    vnd::SpecTable t;
    std::cin >> t.C;
    t.tap_off.push_back(0);
    for (int c = 0; c < t.C; ++c) {
        int k; std::cin >> k;
        for (int i = 0; i < k; ++i) { int v; std::cin >> v; t.idx.push_back(v); t.max_index = std::max(t.max_index, v); }
        for (int i = 0; i < k; ++i) { float v; std::cin >> v; t.w.push_back(v); }
        t.tap_off.push_back((int)t.idx.size());
    }
    vnd::SpecConfig cfg;
    if (!vnd::spec_pick_config(t, 160 * 1024, argc > 1 ? atoi(argv[1]) : 0, argc > 2 ? atoi(argv[2]) : 0, &cfg)) { fprintf(stderr, "no config\n"); return 1; }
    if (argc > 3) cfg.la = atoi(argv[3]);
    fprintf(stderr, "rr=%d pp=%d dd=%d la=%d lds=%zu\n", cfg.rr, cfg.pp, cfg.dd, cfg.la, cfg.lds_bytes());
    std::string src = vnd::spec_prologue(t, cfg) + vnd::kSpecKernelSource;
    fputs(src.c_str(), stdout);
    return 0;
}
