// prof.h - kernel-class ids for the optional event timer (prof.hip)
#pragma once
#include <hip/hip_runtime.h>
enum {
    EH_K_DEGREE = 0, EH_K_SELECT, EH_K_OFFSETS, EH_K_TIME_EMBED, EH_K_NODE_IN, EH_K_GRAPH_UPDATE, EH_K_GEMM_QT,
    EH_K_ATTN, EH_K_GEMM_OUT, EH_K_GRU, EH_K_CTX, EH_K_GATHER, EH_K_GEMM_TAIL, EH_K_PAIR, EH_K_QSAMPLE, EH_K_LAYER, EH_K_MISC, EH_K_TRAIN,
    EH_K_COUNT
};
bool eh_prof_on();
void eh_prof_begin(hipStream_t s);
void eh_prof_end(int id, hipStream_t s);
// wrap one launch:  EH_PROF(id, stream, hipLaunchKernelGGL(...));
#define EH_PROF(id, s, launch) do { eh_prof_begin(s); launch; eh_prof_end(id, s); } while (0)
