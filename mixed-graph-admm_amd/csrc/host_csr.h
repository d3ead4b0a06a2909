// Host-side CSR container (plain C++: also used by the CPU-only checks of the tile builders).
#pragma once
#include <vector>

struct HostCsr {
    int n = 0;
    std::vector<int> rowptr, col;
    std::vector<float> val;
    int nnz() const { return (int)col.size(); }
};
