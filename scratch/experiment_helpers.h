// helpers the patches under scratch/ (wave_kernels_experiment, latency_fixes_experiment, bwd_prefetch_experiment) expect in
// gru_layout.h / mma_core.h; removed from the product headers because nothing shipped uses them

// The same through the scalar unit (wave-uniform address): the answer comes back on lgkmcnt, so a poll does not have to wait for the
// vector stores / gathers the wavefront still has in flight (vmcnt returns in order).  glc: served by L2, where the counter's atomics execute.
FN_DEVINL const u32* fn_uniform(const u32* p) {       // a wave-uniform pointer in scalar registers
    const unsigned long long v = (unsigned long long)p;
    const u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
    return reinterpret_cast<const u32*>(((unsigned long long)hi << 32) | lo);
}
FN_DEVINL u32 ld_cnt_s(const u32* p) {
    u32 v;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// Workgroup barrier that only settles LDS traffic.  __syncthreads() is a fence + barrier: hipcc drains vmcnt(0) in front of it, i.e. every
// barrier of the time loop would wait for the gathers / operand prefetches a step has in flight on purpose.  Global hand-overs
// are ordered explicitly where they happen (s_waitcnt vmcnt(0) before the arrival atomic).
FN_DEVINL void fn_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// the same with a count that is only known after unrolling (folds to one s_waitcnt)
FN_DEVINL void fn_wait_vm_n(int n) {
    switch (n) {
#define FN_WVM_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        FN_WVM_CASE(0) FN_WVM_CASE(1) FN_WVM_CASE(2) FN_WVM_CASE(3) FN_WVM_CASE(4) FN_WVM_CASE(5) FN_WVM_CASE(6) FN_WVM_CASE(7)
        FN_WVM_CASE(8) FN_WVM_CASE(9) FN_WVM_CASE(10) FN_WVM_CASE(11) FN_WVM_CASE(12) FN_WVM_CASE(13) FN_WVM_CASE(14) FN_WVM_CASE(15)
        FN_WVM_CASE(16) FN_WVM_CASE(17) FN_WVM_CASE(18) FN_WVM_CASE(19) FN_WVM_CASE(20) FN_WVM_CASE(21) FN_WVM_CASE(22) FN_WVM_CASE(23)
        FN_WVM_CASE(24) FN_WVM_CASE(25) FN_WVM_CASE(26) FN_WVM_CASE(27) FN_WVM_CASE(28) FN_WVM_CASE(29) FN_WVM_CASE(30) FN_WVM_CASE(31)
#undef FN_WVM_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    __builtin_amdgcn_sched_barrier(0);
}

