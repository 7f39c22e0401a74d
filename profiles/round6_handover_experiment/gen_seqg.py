"""Generate plan_bits_seqg_kernel + seq_run_pose (sixteen poses per workgroup, hand-overs inside the CU) from plan_bits_seq_kernel's text."""
p='quadrupedal_foothold_planner_amd/csrc/fpe_bits.hpp'
s=open(p).read()
assert 'plan_bits_seqg_kernel' not in s
start=s.index('template <int NRL, int KW, int kProd>\n__global__ __launch_bounds__(64, FPE_SEQ_WAVES) void plan_bits_seq_kernel(')
end=s.index('// ---- host side of the bit-window path')
k=s[start:end]
def rep(old,new,cnt=1):
    global k
    assert k.count(old)==cnt, (k.count(old), old[:80])
    k=k.replace(old,new)
# ---------- the callee: from the kernel's text
rep('''template <int NRL, int KW, int kProd>
__global__ __launch_bounds__(64, FPE_SEQ_WAVES) void plan_bits_seq_kernel(DevMap m, BitMap bm, PlanConsts pc, SpiralLut lut,
                                                                          const fpe_pose* __restrict__ poses, int B, int nCycles, fpe_plan_out outArg, int recSlots) {
    constexpr int G = 64;
    const fpe_plan_out out = specialise_products<kProd>(outArg);
    constexpr int NR = G * NRL;
    stamp(pc, 6, 14);  // (profiling builds: lifetime of the wavefront, with the stamp after the cycle loop)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = static_cast<int>(threadIdx.x);
    const Grp<G> g(tid);
    PoseShared& sh = *reinterpret_cast<PoseShared*>(smem);
#ifdef FPE_TRACE
    if (tid < 4) sh.pad[tid] = 0;
#endif
    // per-leg constants of the pose, computed once (lane = leg) instead of once per leg and phase: a division and a
    // dependent rank-table load each
    LegStatic* lsTab = reinterpret_cast<LegStatic*>(smem + sizeof(PoseShared));
    constexpr size_t kLsBytes = (4 * sizeof(LegStatic) + 15) & ~static_cast<size_t>(15);
    // rows actually allocated: the window's 2 winH + 1 (not 64 * NRL) — LDS bounds the occupancy of these kernels
    const LegBits lb = make_legbits(smem + sizeof(PoseShared) + kLsBytes, min(2 * pc.winH + 1, NR), KW, pc.nHW, true);
    // staged output records: recSlots (a power of two, sized by the launch to keep the LDS within the occupancy budget)
    // cycles of four legs behind the row arrays
    using Rec = SeqRecOf<KW>;
    Rec* recBase = reinterpret_cast<Rec*>(
        smem + ((sizeof(PoseShared) + kLsBytes + 4 * static_cast<size_t>(legbits_words(min(2 * pc.winH + 1, NR), KW, pc.nHW, true)) + 15) & ~static_cast<size_t>(15)));
    const int b = blockIdx.x;
    if (b >= B) return;
    const bool live = true;

    const fpe_pose* pp = poses + b;
    const double x0 = pp->position[0], y0 = pp->position[1], z0 = pp->position[2];
    const int gait = pp->gait;
    const LutHead head = load_lut_head(lut, g);
''','''// One pose's chain from gait cycle cyc0 (fresh: from its stance) until its last cycle or until it is handed to another wavefront of
// the workgroup: a FUNCTION, not inlined into the kernel's loop over poses — inlined, the loop carried forty more vector registers
// than the 128 the four-wavefront budget allows (round 4 saw the same: +36 VGPRs in scratch).  Everything uniform is read from the
// kernel's argument segment (`kaArg`) or from the pose's LDS slot.  Returns true when the pose was given away.
template <int NRL, int KW, int kProd>
__device__ __attribute__((noinline)) bool seq_run_pose(const SeqKernArgs __attribute__((address_space(4))) * kaIn, int slotBytesIn, int slotIn, int bInV,
                                                       bool freshIn, int tid, int simdIn, unsigned hwidIn, const LutHead& head) {
    constexpr int G = 64;
    constexpr int NR = G * NRL;
    // (a function's arguments arrive in VECTOR registers: the uniform ones go back to scalars here, or every address and index derived
    // from them would be vector arithmetic — and the argument-segment pointer could not feed scalar loads at all)
    typedef const SeqKernArgs __attribute__((address_space(4))) * KernArgPtrS;
    const unsigned long long kaBits = reinterpret_cast<unsigned long long>(kaIn);
    const KernArgPtrS kaArg = reinterpret_cast<KernArgPtrS>((static_cast<unsigned long long>(static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(kaBits >> 32)))) << 32) |
                                                              static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(kaBits))));
    const int slotBytes = __builtin_amdgcn_readfirstlane(slotBytesIn), slot = __builtin_amdgcn_readfirstlane(slotIn), bIn = __builtin_amdgcn_readfirstlane(bInV);
    const bool fresh = __builtin_amdgcn_readfirstlane(freshIn ? 1 : 0) != 0;
    const int simd = __builtin_amdgcn_readfirstlane(simdIn);
    const unsigned hwid = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(hwidIn)));
    // (the workgroup's LDS by its own symbol: a pointer PARAMETER would be a generic one, and every LDS access a flat instruction)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const SeqKernArgs* kaG = (const SeqKernArgs*)kaArg;
    const DevMap& m = kaG->m;
    const BitMap& bm = kaG->bm;
    const PlanConsts& pc = kaG->pc;
    const SpiralLut& lut = kaG->lut;
    const fpe_pose* __restrict__ poses = kaG->poses;
    const int nCycles = kaG->nCycles, recSlots = kaG->recSlots;
    const fpe_plan_out out = specialise_products<kProd>(kaG->out);
    const Grp<G> g(tid);
    SeqSteal& st = *reinterpret_cast<SeqSteal*>(smem);
    const bool live = true;
    constexpr size_t kLsBytes = (4 * sizeof(LegStatic) + 15) & ~static_cast<size_t>(15);
    using Rec = SeqRecOf<KW>;
    int b = bIn;
    unsigned char* const pbase = smem + sizeof(SeqSteal) + static_cast<size_t>(slot) * slotBytes;
    PoseShared& sh = *reinterpret_cast<PoseShared*>(pbase);
    LegStatic* lsTab = reinterpret_cast<LegStatic*>(pbase + sizeof(PoseShared));
    const LegBits lb = make_legbits(pbase + sizeof(PoseShared) + kLsBytes, min(2 * pc.winH + 1, NR), KW, pc.nHW, true);
    Rec* recBase = reinterpret_cast<Rec*>(
        pbase + ((sizeof(PoseShared) + kLsBytes + 4 * static_cast<size_t>(legbits_words(min(2 * pc.winH + 1, NR), KW, pc.nHW, true)) + 15) & ~static_cast<size_t>(15)));
    SeqCarry& carry = *reinterpret_cast<SeqCarry*>(pbase + slotBytes - sizeof(SeqCarry));
    double y0, adjY = 0.0;  // ajustedPose_[1], cpp:759
    int gait, cyc0 = 0, ph0 = 0;
    bool cycleOk0 = true;
    if (fresh) {
#ifdef FPE_TRACE
    if (tid < 4) sh.pad[tid] = 0;
#endif
    const fpe_pose* pp = poses + b;
    const double x0 = pp->position[0], z0 = pp->position[2];
    y0 = pp->position[1];
    gait = pp->gait;
''')
rep('''    pose_sync<16>();
    if (out.pose_status && tid == 0) out.pose_status[b] = opt_gate_cycle0(m.g, pc, polygon_center_x(sh.cur[0]), y0);

    double adjY = 0.0;  // ajustedPose_[1], cpp:759
    const int nPhases''','''    pose_sync<16>();
    if (out.pose_status && tid == 0) out.pose_status[b] = opt_gate_cycle0(m.g, pc, polygon_center_x(sh.cur[0]), y0);
    } else {  // a pose taken over at the end of one of its gait cycles: everything else of it lives in the slot's LDS
        b = __builtin_amdgcn_readfirstlane(carry.b);
        cyc0 = __builtin_amdgcn_readfirstlane(carry.cyc);
        gait = __builtin_amdgcn_readfirstlane(carry.gait);
        y0 = carry.y0;
        adjY = carry.adjY;
        ph0 = __builtin_amdgcn_readfirstlane(carry.ph);
        cycleOk0 = __builtin_amdgcn_readfirstlane(carry.cycleOk) != 0;
    }
    const int nPhases''')
rep('''    int cycLag;  // launch order of this wavefront on its SIMD (HW_ID.WAVE_ID: 0 oldest .. 3) x a sixteenth of the cycles
    {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        cycLag = (static_cast<int>(hwid & 3u) * nCycles) / 16;
    }
''','''    const int cycLag = (static_cast<int>(hwid & 3u) * nCycles) / 16;  // launch order of this wavefront on its SIMD x a sixteenth of the cycles
    bool gave = false;
''')
rep('''    for (int cyc = 0; cyc < nCycles; ++cyc) {
        {
            // Issue priority by PROGRESS''','''    for (int cyc = cyc0; cyc < nCycles; ++cyc) {
        {
            // Issue priority by PROGRESS''')
rep('''        bool cycleOk = true;
        for (int ph = 0; ph < nPhases; ++ph) {''','''        bool cycleOk = cyc == cyc0 ? cycleOk0 : true;
        for (int ph = (cyc == cyc0 ? ph0 : 0); ph < nPhases; ++ph) {''')
rep('''            cycleOk = cycleOk && phaseOk;
            stamp(pc, cyc, 10);
        }''','''            cycleOk = cycleOk && phaseOk;
            stamp(pc, cyc, 10);
            // the end of a phase inside a gait cycle (walk gait): see the end of the cycle below
            if (ph + 1 < nPhases && seq_ld(&st.idle) > 0) {
                int target = -1;
                if (tid == 0) target = seq_try_give(st, simd);
                target = __builtin_amdgcn_readfirstlane(target);
                if (target >= 0) {
                    if (tid == 0) {
                        carry.b = b; carry.cyc = cyc; carry.gait = gait; carry.y0 = y0; carry.adjY = adjY;
                        carry.ph = ph + 1; carry.cycleOk = cycleOk ? 1 : 0;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (tid == 0) {
                        seq_add(&st.active[seq_ld(&st.simdOf[target])], 1);
                        __hip_atomic_store(&st.box[target], slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    gave = true;
                    break;
                }
            }
        }
        if (gave) break;''')
rep('make_legbits(smem + sizeof(PoseShared) + kLsBytes, rowsL','make_legbits(pbase + sizeof(PoseShared) + kLsBytes, rowsL')
rep('''                    smem + ((sizeof(PoseShared) + kLsBytes + 4 * static_cast<size_t>(legbits_words(rowsL''','''                    pbase + ((sizeof(PoseShared) + kLsBytes + 4 * static_cast<size_t>(legbits_words(rowsL''')
rep('''reinterpret_cast<Rec*>(smem + ((sizeof(PoseShared) + kLsBytes +''','''reinterpret_cast<Rec*>(pbase + ((sizeof(PoseShared) + kLsBytes +''')
assert 'smem + (' not in k.replace('smem + sizeof(SeqSteal)','')
rep('''#ifdef FPE_TRACE
                __builtin_amdgcn_s_waitcnt(0);
                flushClocks += static_cast<long long>(__builtin_readcyclecounter()) - tFlush0;
                ++nFlushes;
#endif
            }
        }
    }
    stamp(pc, 6, 15);''','''#ifdef FPE_TRACE
                __builtin_amdgcn_s_waitcnt(0);
                flushClocks += static_cast<long long>(__builtin_readcyclecounter()) - tFlush0;
                ++nFlushes;
#endif
            }
        }
        // the end of a gait cycle that is not the pose's last: a SIMD of this CU with at least two poses fewer than this one, and a
        // wavefront waiting on it?  Then the pose continues THERE (its state is this slot's LDS plus five words)
        if (cyc + 1 < nCycles && seq_ld(&st.idle) > 0) {
            int target = -1;
            if (tid == 0) target = seq_try_give(st, simd);
            target = __builtin_amdgcn_readfirstlane(target);
            if (target >= 0) {
                if (tid == 0) {
                    carry.b = b;
                    carry.cyc = cyc + 1;
                    carry.gait = gait;
                    carry.y0 = y0;
                    carry.adjY = adjY;
                    carry.ph = 0;
                    carry.cycleOk = 1;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (tid == 0) {
                    seq_add(&st.active[seq_ld(&st.simdOf[target])], 1);
                    __hip_atomic_store(&st.box[target], slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                gave = true;
                break;
            }
        }
    }
    if (!gave && tid == 0) {  // the pose has run its last cycle
        seq_add(&st.active[simd], -1);
        seq_add(&st.remaining, -1);
    }
    stamp(pc, 6, 15);''')
assert k.count('(KernArgPtr)__builtin_amdgcn_kernarg_segment_ptr()')==2
k=k.replace('(KernArgPtr)__builtin_amdgcn_kernarg_segment_ptr()','(KernArgPtr)kaArg')
k=k.rstrip()
assert k.endswith('}')
k=k[:-1]+'    return gave;\n}\n\n'
kernel='''template <int NRL, int KW, int kProd>
__global__ __launch_bounds__(64 * kSeqGroup) void plan_bits_seqg_kernel(DevMap m, BitMap bm, PlanConsts pc, SpiralLut lut,
                                                                        const fpe_pose* __restrict__ poses, int B, int nCycles, fpe_plan_out outArg, int recSlots, int slotBytes) {
    constexpr int G = 64;
    stamp(pc, 6, 14);  // (profiling builds: lifetime of the wavefront)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = static_cast<int>(threadIdx.x) & 63, wv = static_cast<int>(threadIdx.x) >> 6;
    const Grp<G> g(tid);
    SeqSteal& st = *reinterpret_cast<SeqSteal*>(smem);
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const int simd = static_cast<int>((hwid >> 4) & 3u);
    if (threadIdx.x < 4) st.active[threadIdx.x] = 0;
    if (threadIdx.x == 4) st.remaining = st.idle = 0;
    if (threadIdx.x < kSeqGroup) {
        st.state[threadIdx.x] = 0;
        st.box[threadIdx.x] = 0;
    }
    __syncthreads();
    int slot = wv;
    const int b = blockIdx.x * kSeqGroup + slot;
    bool have = b < B;
    if (tid == 0) {
        st.simdOf[wv] = simd;
        if (have) {
            seq_add(&st.active[simd], 1);
            seq_add(&st.remaining, 1);
        }
    }
    __syncthreads();  // (the only workgroup barriers of the kernel: from here on the wavefronts never meet again)
    const LutHead head = load_lut_head(lut, g);
    bool fresh = true;
    (void)outArg; (void)nCycles; (void)recSlots; (void)poses; (void)m; (void)bm;
    typedef const SeqKernArgs __attribute__((address_space(4))) * KernArgPtr0;
    for (;;) {  // the poses this wavefront runs: its own, then whatever an overloaded SIMD of the CU hands to it
        if (have) (void)seq_run_pose<NRL, KW, kProd>((KernArgPtr0)__builtin_amdgcn_kernarg_segment_ptr(), slotBytes, slot, b, fresh, tid, simd, hwid, head);
        // a wavefront without a pose waits for one (LDS polls between sleeps) until every pose of the workgroup has ended
        __builtin_amdgcn_s_setprio(0);
        int got = -1;
        if (tid == 0) got = seq_take(st, wv);
        got = __builtin_amdgcn_readfirstlane(got);
        if (got < 0) break;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        slot = got;
        have = true;
        fresh = false;
    }
}

'''
helpers=open('scratch/seqg_helpers.txt').read()
s=s[:end]+helpers+k+kernel+s[end:]
old='''#ifndef FPE_SEQ_RELOAD_ARGS
#define FPE_SEQ_RELOAD_ARGS 2
#endif'''
new=old+'''
#ifndef FPE_SEQ_GROUP  // 1: batches of >= 64 poses run sixteen poses per workgroup with hand-overs inside the CU (plan_bits_seqg_kernel)
#define FPE_SEQ_GROUP 0
#endif'''
assert old in s
s=s.replace(old,new)
old='''        int recSlots = 8; /* cycles of staged records: as many as keep sixteen blocks per CU (10 KiB each) */                      \\
        while (recSlots > 1 && base + recSlots * 4 * sizeof(SeqRecOf<KW>) > 10240) recSlots >>= 1;                                   \\
'''
assert old in s
s=s.replace(old,old+open('scratch/seqg_launch.txt').read())
open(p,'w').write(s)
print('ok')
