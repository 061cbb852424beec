// Micro-benchmark: issue cost of the Hamming row's instructions in MEASURED shader cycles.
//
// Round 2's class_order.hip converted wall time into cycles with an ASSUMED 2.4 GHz clock.  This one stamps
// s_memtime (shader cycles) and s_memrealtime (100 MHz constant clock) around the loop in every wave, so it reports
//   * the clock the loop really ran at (delta s_memtime / delta s_memrealtime x 100 MHz, median over waves),
//   * cycles per instruction per SIMD = launch span (first to last stamp) x that clock / instructions per SIMD, and
//   * one wave's own issue cadence (its span / its instructions),
// after >= 2 s of back-to-back launches (MI355X_MICROARCH.md "DVFS give-back" item 6).  Stamps go to a buffer of
// their own.  Bodies: xor only, bcnt only, the shipped row (8 v_xor then 8 v_bcnt), the row + one v_and (the
// filter's share), the alternating order, and the row with two LDS broadcast reads per row as in the kernel.
// build: hipcc --offload-arch=gfx950 -O3 cycles.hip -o cycles
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
#define CLOB "v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47"

#define XOR8 "v_xor_b32 v32, v16, v26\n v_xor_b32 v33, v17, v27\n v_xor_b32 v34, v18, v24\n v_xor_b32 v35, v19, v25\n v_xor_b32 v36, v20, v30\n v_xor_b32 v37, v21, v31\n v_xor_b32 v38, v22, v28\n v_xor_b32 v39, v23, v29\n"
#define BCNT8 "v_bcnt_u32_b32 v40, v32, v41\n v_bcnt_u32_b32 v40, v33, v40\n v_bcnt_u32_b32 v40, v34, v40\n v_bcnt_u32_b32 v40, v35, v40\n v_bcnt_u32_b32 v40, v36, v40\n v_bcnt_u32_b32 v40, v37, v40\n v_bcnt_u32_b32 v40, v38, v40\n v_bcnt_u32_b32 v40, v39, v40\n"
#define ALT8 "v_xor_b32 v32, v16, v26\n v_bcnt_u32_b32 v40, v32, v41\n v_xor_b32 v33, v17, v27\n v_bcnt_u32_b32 v40, v33, v40\n v_xor_b32 v34, v18, v24\n v_bcnt_u32_b32 v40, v34, v40\n v_xor_b32 v35, v19, v25\n v_bcnt_u32_b32 v40, v35, v40\n v_xor_b32 v36, v20, v30\n v_bcnt_u32_b32 v40, v36, v40\n v_xor_b32 v37, v21, v31\n v_bcnt_u32_b32 v40, v37, v40\n v_xor_b32 v38, v22, v28\n v_bcnt_u32_b32 v40, v38, v40\n v_xor_b32 v39, v23, v29\n v_bcnt_u32_b32 v40, v39, v40\n"
#define BCNTONLY8 "v_bcnt_u32_b32 v40, v32, v40\n v_bcnt_u32_b32 v41, v33, v41\n v_bcnt_u32_b32 v42, v34, v42\n v_bcnt_u32_b32 v43, v35, v43\n v_bcnt_u32_b32 v40, v36, v40\n v_bcnt_u32_b32 v41, v37, v41\n v_bcnt_u32_b32 v42, v38, v42\n v_bcnt_u32_b32 v43, v39, v43\n"
#define AND1 "v_and_b32 v42, v42, v40\n"
// the same v_xor in the 8-byte VOP3 encoding (_e64): same operation, twice the instruction bytes - an instruction-fetch probe
#define XOR8_E64 "v_xor_b32_e64 v32, v16, v26\n v_xor_b32_e64 v33, v17, v27\n v_xor_b32_e64 v34, v18, v24\n v_xor_b32_e64 v35, v19, v25\n v_xor_b32_e64 v36, v20, v30\n v_xor_b32_e64 v37, v21, v31\n v_xor_b32_e64 v38, v22, v28\n v_xor_b32_e64 v39, v23, v29\n"
// a 4-cycle-class op in its 4-byte (e32) and 8-byte (e64) encodings
#define MIN8_E32 "v_min_u32_e32 v32, v16, v26\n v_min_u32_e32 v33, v17, v27\n v_min_u32_e32 v34, v18, v24\n v_min_u32_e32 v35, v19, v25\n v_min_u32_e32 v36, v20, v30\n v_min_u32_e32 v37, v21, v31\n v_min_u32_e32 v38, v22, v28\n v_min_u32_e32 v39, v23, v29\n"
#define MIN8_E64 "v_min_u32_e64 v32, v16, v26\n v_min_u32_e64 v33, v17, v27\n v_min_u32_e64 v34, v18, v24\n v_min_u32_e64 v35, v19, v25\n v_min_u32_e64 v36, v20, v30\n v_min_u32_e64 v37, v21, v31\n v_min_u32_e64 v38, v22, v28\n v_min_u32_e64 v39, v23, v29\n"
#define X4(a) a a a a
// v_xor with the train word in an SGPR (wave-uniform operand, no LDS read, one VGPR read less per instruction)
#define XOR8_S "v_xor_b32 v32, s20, v16\n v_xor_b32 v33, s21, v17\n v_xor_b32 v34, s22, v18\n v_xor_b32 v35, s23, v19\n v_xor_b32 v36, s24, v20\n v_xor_b32 v37, s25, v21\n v_xor_b32 v38, s26, v22\n v_xor_b32 v39, s27, v23\n"
#define SCLOB "s20","s21","s22","s23","s24","s25","s26","s27"
// dependency probes
#define BCNT8_SERIAL "v_bcnt_u32_b32 v40, v32, v40\n v_bcnt_u32_b32 v40, v33, v40\n v_bcnt_u32_b32 v40, v34, v40\n v_bcnt_u32_b32 v40, v35, v40\n v_bcnt_u32_b32 v40, v36, v40\n v_bcnt_u32_b32 v40, v37, v40\n v_bcnt_u32_b32 v40, v38, v40\n v_bcnt_u32_b32 v40, v39, v40\n"
// row with two accumulator chains (even / odd words) joined by one v_add
#define BCNT8_2CH "v_bcnt_u32_b32 v40, v32, v41\n v_bcnt_u32_b32 v43, v33, v42\n v_bcnt_u32_b32 v40, v34, v40\n v_bcnt_u32_b32 v43, v35, v43\n v_bcnt_u32_b32 v40, v36, v40\n v_bcnt_u32_b32 v43, v37, v43\n v_bcnt_u32_b32 v40, v38, v40\n v_bcnt_u32_b32 v43, v39, v43\n v_add_u32 v40, v40, v43\n"
// two rows at once: 16 xor (second row into v12-v15, v44-v47), then the two rows' chains alternate
#define XOR8_B "v_xor_b32 v12, v16, v26\n v_xor_b32 v13, v17, v27\n v_xor_b32 v14, v18, v24\n v_xor_b32 v15, v19, v25\n v_xor_b32 v44, v20, v30\n v_xor_b32 v45, v21, v31\n v_xor_b32 v46, v22, v28\n v_xor_b32 v47, v23, v29\n"
#define BCNT16_2ROWS "v_bcnt_u32_b32 v40, v32, v41\n v_bcnt_u32_b32 v43, v12, v41\n v_bcnt_u32_b32 v40, v33, v40\n v_bcnt_u32_b32 v43, v13, v43\n v_bcnt_u32_b32 v40, v34, v40\n v_bcnt_u32_b32 v43, v14, v43\n v_bcnt_u32_b32 v40, v35, v40\n v_bcnt_u32_b32 v43, v15, v43\n v_bcnt_u32_b32 v40, v36, v40\n v_bcnt_u32_b32 v43, v44, v43\n v_bcnt_u32_b32 v40, v37, v40\n v_bcnt_u32_b32 v43, v45, v43\n v_bcnt_u32_b32 v40, v38, v40\n v_bcnt_u32_b32 v43, v46, v43\n v_bcnt_u32_b32 v40, v39, v40\n v_bcnt_u32_b32 v43, v47, v43\n"
// xor into FRESH registers that no recent bcnt read (two register sets alternate): removes the write-after-read on the row buffer
#define ROW_A XOR8 BCNT8
#define ROW_B XOR8_B "v_bcnt_u32_b32 v40, v12, v41\n v_bcnt_u32_b32 v40, v13, v40\n v_bcnt_u32_b32 v40, v14, v40\n v_bcnt_u32_b32 v40, v15, v40\n v_bcnt_u32_b32 v40, v44, v40\n v_bcnt_u32_b32 v40, v45, v40\n v_bcnt_u32_b32 v40, v46, v40\n v_bcnt_u32_b32 v40, v47, v40\n"

struct stamp { unsigned long long c0, r0, c1, r1; };

// MODE 0 xor x32 | 1 bcnt x32 | 2 row x2 (8 xor, 8 bcnt) | 3 (row + and) x2 | 4 alternating x2 | 5 row x2 with 2 LDS b128 reads per row
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, stamp* st, int iters) {
    __shared__ uint4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_uint4(i, i * 3, i * 5, i * 7);
    __syncthreads();
    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x;
    asm volatile("v_mov_b32 v12, %0\nv_mov_b32 v13, %0\nv_mov_b32 v14, %0\nv_mov_b32 v15, %0\nv_mov_b32 v16, %0\nv_mov_b32 v17, %0\nv_mov_b32 v18, %0\nv_mov_b32 v19, %0\nv_mov_b32 v20, %0\nv_mov_b32 v21, %0\nv_mov_b32 v22, %0\nv_mov_b32 v23, %0\nv_mov_b32 v24, %0\nv_mov_b32 v25, %0\nv_mov_b32 v26, %0\nv_mov_b32 v27, %0\nv_mov_b32 v28, %0\nv_mov_b32 v29, %0\nv_mov_b32 v30, %0\nv_mov_b32 v31, %0\nv_mov_b32 v32, %0\nv_mov_b32 v33, %0\nv_mov_b32 v34, %0\nv_mov_b32 v35, %0\nv_mov_b32 v36, %0\nv_mov_b32 v37, %0\nv_mov_b32 v38, %0\nv_mov_b32 v39, %0\nv_mov_b32 v40, %0\nv_mov_b32 v41, %0\nv_mov_b32 v42, %0\nv_mov_b32 v43, %0\nv_mov_b32 v44, %0\nv_mov_b32 v45, %0\nv_mov_b32 v46, %0\nv_mov_b32 v47, %0" ::"v"(seed) : CLOB);
    unsigned long long c0, r0, c1, r1;
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const bool slot_odd = hwid & 1;
    asm volatile("s_mov_b32 s20, 0x12345678\n s_mov_b32 s21, 0x9abcdef0\n s_mov_b32 s22, 0x0f1e2d3c\n s_mov_b32 s23, 0x4b5a6978\n"
                 "s_mov_b32 s24, 0x87969fa5\n s_mov_b32 s25, 0xb4c3d2e1\n s_mov_b32 s26, 0x13579bdf\n s_mov_b32 s27, 0x2468ace0" ::: SCLOB);
    if ((MODE == 21 || MODE == 32) && slot_odd) asm volatile("s_setprio 3");
    if (MODE == 22 && !slot_odd) asm volatile("s_setprio 3");
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    uint32_t addr = 0;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) asm volatile(XOR8 XOR8 XOR8 XOR8 ::: CLOB);
        if (MODE == 1) asm volatile(BCNTONLY8 BCNTONLY8 BCNTONLY8 BCNTONLY8 ::: CLOB);
        if (MODE == 2) asm volatile(XOR8 BCNT8 XOR8 BCNT8 ::: CLOB);
        if (MODE == 3) asm volatile(XOR8 BCNT8 AND1 XOR8 BCNT8 AND1 ::: CLOB);
        if (MODE == 4) asm volatile(ALT8 ALT8 ::: CLOB);
        if (MODE == 6) asm volatile(XOR8_E64 XOR8_E64 XOR8_E64 XOR8_E64 ::: CLOB);
        if (MODE == 7) { asm volatile(X4(X4(XOR8 XOR8)) ::: CLOB); it += 7; }            // 256 instructions per loop iteration
        if (MODE == 8) { asm volatile(X4(X4(XOR8 BCNT8)) ::: CLOB); it += 7; }           // 16 rows per loop iteration
        if (MODE == 9) asm volatile(MIN8_E32 MIN8_E32 MIN8_E32 MIN8_E32 ::: CLOB);
        if (MODE == 10) asm volatile(MIN8_E64 MIN8_E64 MIN8_E64 MIN8_E64 ::: CLOB);
        if (MODE == 11) { asm volatile(X4(X4(BCNTONLY8 BCNTONLY8)) ::: CLOB); it += 7; }
        if (MODE == 12) { asm volatile(X4(X4(BCNT8_SERIAL BCNT8_SERIAL)) ::: CLOB); it += 7; }
        if (MODE == 13) { asm volatile(X4(X4(XOR8 BCNT8_2CH)) ::: CLOB); it += 7; }               // 17 instructions per row
        if (MODE == 14) { asm volatile(X4(XOR8 XOR8_B BCNT16_2ROWS XOR8 XOR8_B BCNT16_2ROWS) ::: CLOB); it += 7; }
        if (MODE == 15) { asm volatile(X4(ROW_A ROW_B ROW_A ROW_B) ::: CLOB); it += 7; }
        if (MODE == 16) { asm volatile(X4(X4(ALT8)) ::: CLOB); it += 7; }
        if (MODE >= 20 && MODE <= 23) {
            // co-issue probe: the waves in even wave slots of a SIMD run the pure xor stream, those in odd slots the pure
            // bcnt stream (HW_ID[3:0] = wave slot).  MODE 21: the bcnt waves run at s_setprio 3; 22: the xor waves do; 23: no
            // priority, but the xor waves run TWICE the instructions (equal ALU time per wave if xor = 2 and bcnt = 4 cycles)
            if (slot_odd) { asm volatile(X4(X4(BCNTONLY8 BCNTONLY8)) ::: CLOB); }
            else { asm volatile(X4(X4(XOR8 XOR8)) ::: CLOB); if (MODE == 23) asm volatile(X4(X4(XOR8 XOR8)) ::: CLOB); }
            it += 7;
        }
        // the row stream with the wave's priority raised for its bcnt phase (s_setprio is free-standing, SOPP)
        if (MODE == 24) { asm volatile(X4(X4("s_setprio 0\n" XOR8 "s_setprio 2\n" BCNT8)) ::: CLOB); it += 7; }
        if (MODE == 25) { asm volatile(X4("s_setprio 0\n" XOR8 XOR8_B "s_setprio 2\n" BCNT16_2ROWS "s_setprio 0\n" XOR8 XOR8_B "s_setprio 2\n" BCNT16_2ROWS) ::: CLOB); it += 7; }
        if (MODE == 26) { asm volatile(X4(X4("s_setprio 2\n" XOR8 "s_setprio 0\n" BCNT8)) ::: CLOB); it += 7; }
        if (MODE == 27) { asm volatile(X4(X4("s_setprio 0\n" XOR8 "s_setprio 3\n" BCNT8 AND1)) ::: CLOB); it += 7; }
        if (MODE == 28 || MODE == 29 || MODE == 30) {
            // the kernel's row with priorities: reads of the NEXT row are issued first (rolling, two 8-register buffers:
            // v24-31 / v12-15,v44-47 would need renaming per row, so here both rows xor out of place into v32-39, which keeps
            // the instruction mix and the LDS traffic of the kernel: 2 ds_read_b128 per 16 (MODE 28), per 32 (MODE 29: two
            // queries per lane) VALU instructions; MODE 30 = MODE 28 without the reads (same waits removed)
            if (MODE == 28)
                asm volatile(X4("ds_read_b128 v[24:27], %0\n ds_read_b128 v[28:31], %0 offset:16\n s_waitcnt lgkmcnt(2)\n"
                                "s_setprio 0\n" XOR8 "s_setprio 2\n" BCNT8 AND1
                                "ds_read_b128 v[24:27], %0 offset:32\n ds_read_b128 v[28:31], %0 offset:48\n s_waitcnt lgkmcnt(2)\n"
                                "s_setprio 0\n" XOR8 "s_setprio 2\n" BCNT8 AND1
                                "ds_read_b128 v[24:27], %0 offset:64\n ds_read_b128 v[28:31], %0 offset:80\n s_waitcnt lgkmcnt(2)\n"
                                "s_setprio 0\n" XOR8 "s_setprio 2\n" BCNT8 AND1
                                "ds_read_b128 v[24:27], %0 offset:96\n ds_read_b128 v[28:31], %0 offset:112\n s_waitcnt lgkmcnt(2)\n"
                                "s_setprio 0\n" XOR8 "s_setprio 2\n" BCNT8 AND1) ::"v"(addr) : CLOB);
            if (MODE == 29)
                asm volatile(X4("ds_read_b128 v[24:27], %0\n ds_read_b128 v[28:31], %0 offset:16\n s_waitcnt lgkmcnt(2)\n"
                                "s_setprio 0\n" XOR8 XOR8_B "s_setprio 2\n" BCNT16_2ROWS AND1 AND1
                                "ds_read_b128 v[24:27], %0 offset:32\n ds_read_b128 v[28:31], %0 offset:48\n s_waitcnt lgkmcnt(2)\n"
                                "s_setprio 0\n" XOR8 XOR8_B "s_setprio 2\n" BCNT16_2ROWS AND1 AND1) ::"v"(addr) : CLOB);
            if (MODE == 30)
                asm volatile(X4(X4("s_setprio 0\n" XOR8 "s_setprio 2\n" BCNT8 AND1)) ::: CLOB);
            addr = (addr + 128) & 8191;
            it += 7;
        }
        if (MODE == 31) { asm volatile(X4(X4(XOR8_S XOR8_S)) ::: CLOB, SCLOB); it += 7; }
        if (MODE == 32) {
            if (slot_odd) { asm volatile(X4(X4(BCNTONLY8 BCNTONLY8)) ::: CLOB); }
            else { asm volatile(X4(X4(XOR8_S XOR8_S)) ::: CLOB, SCLOB); }
            it += 7;
        }
        if (MODE == 33) { asm volatile(X4(X4("s_setprio 0\n" XOR8_S "s_setprio 2\n" BCNT8)) ::: CLOB, SCLOB); it += 7; }
        if (MODE == 34) { asm volatile(X4(X4("s_setprio 0\n" XOR8_S "s_setprio 2\n" BCNT8 AND1)) ::: CLOB, SCLOB); it += 7; }
        if (MODE == 5) {
            // the kernel's row: two wave-uniform (broadcast) 16-byte LDS reads land in the registers the next row's xors
            // consume; the wait is for the reads issued one row earlier
            asm volatile("ds_read_b128 v[24:27], %0\n ds_read_b128 v[28:31], %0 offset:16\n"
                         XOR8 BCNT8 AND1
                         "s_waitcnt lgkmcnt(0)\n"
                         "ds_read_b128 v[24:27], %0 offset:32\n ds_read_b128 v[28:31], %0 offset:48\n"
                         XOR8 BCNT8 AND1
                         "s_waitcnt lgkmcnt(0)\n" ::"v"(addr) : CLOB);
            addr = (addr + 64) & 8191;
        }
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    uint32_t s;
    asm volatile("v_add_u32 %0, v40, v41\n v_add_u32 %0, %0, v42\n v_add_u32 %0, %0, v43\n v_add_u32 %0, %0, v32" : "=v"(s)::CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        stamp v = {c0, r0, c1, r1};
        st[blockIdx.x * 4 + (threadIdx.x >> 6)] = v;
    }
}

static double median(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <int MODE> int run(const char* name, int ninst, int blocks_per_cu, uint32_t* out, stamp* d_st, double spin_s) {
    const int iters = 20000, ncu = 256, blocks = ncu * blocks_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // spin-up: back-to-back launches of the same body for spin_s seconds
    CK(hipEventRecord(e0));
    float spun = 0;
    while (spun < spin_s * 1e3f) {
        for (int i = 0; i < 8; i++) k<MODE><<<blocks, 256>>>(out, d_st, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&spun, e0, e1));
    }
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, 256>>>(out, d_st, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<stamp> st(blocks * 4);
    CK(hipMemcpy(st.data(), d_st, st.size() * sizeof(stamp), hipMemcpyDeviceToHost));
    std::vector<double> clk, span;
    unsigned long long first = ~0ull, last = 0;
    for (auto& s : st) {
        clk.push_back((double)(s.c1 - s.c0) / (double)(s.r1 - s.r0) * 100.0);                 // MHz
        span.push_back((double)(s.c1 - s.c0) / ((double)iters * ninst));                      // this wave's own cycles per instruction
        first = s.r0 < first ? s.r0 : first;
        last = s.r1 > last ? s.r1 : last;
    }
    const double mclk = median(clk);
    // the SIMD's cost per instruction: (first stamp -> last stamp of the launch, on the 100 MHz clock) x the measured
    // clock / instructions issued per SIMD.  A wave's own span / waves would be wrong: the arbiter serves waves by age,
    // so the waves of one SIMD do not run side by side for the whole launch.
    const double launch_cyc = (double)(last - first) * 1e-8 * mclk * 1e6;
    printf("%-36s waves/SIMD %d: %.3f cycles/instr/SIMD at the measured clock %4.0f MHz (p0 %4.0f, p100 %4.0f) | "
           "one wave's own cadence: median %.2f cycles/instr | event %.3f ms; wall x assumed 2.4 GHz would say %.3f\n",
           name, blocks_per_cu, launch_cyc / ((double)blocks_per_cu * iters * ninst), mclk, clk.front(), clk.back(), median(span), ms,
           ms * 1e-3 * 2.4e9 / ((double)blocks_per_cu * iters * ninst));
    return 0;
}

int main(int argc, char** argv) {
    const double spin_s = argc > 1 ? atof(argv[1]) : 2.0;
    uint32_t* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    stamp* d_st; CK(hipMalloc(&d_st, 256 * 8 * 4 * sizeof(stamp)));
    for (int w : {8, 4}) {
        run<0>("v_xor_b32 x32", 32, w, out, d_st, spin_s);
        run<1>("v_bcnt_u32_b32 x32", 32, w, out, d_st, spin_s);
        run<2>("row: 8 xor then 8 bcnt (x2)", 32, w, out, d_st, spin_s);
        run<3>("row + v_and (x2)", 34, w, out, d_st, spin_s);
        run<4>("alternating xor,bcnt (x2)", 32, w, out, d_st, spin_s);
        run<5>("row + v_and + 2 ds_read_b128 (x2)", 34, w, out, d_st, spin_s);
        run<6>("v_xor_b32_e64 x32 (8-byte encoding)", 32, w, out, d_st, spin_s);
        run<7>("v_xor_b32 x256 per iteration", 32, w, out, d_st, spin_s);
        run<11>("v_bcnt x256 per iteration", 32, w, out, d_st, spin_s);
        run<8>("row x16 per iteration", 32, w, out, d_st, spin_s);
        run<20>("slots: even xor-only, odd bcnt-only", 32, w, out, d_st, spin_s);
        run<21>("  same, bcnt waves s_setprio 3", 32, w, out, d_st, spin_s);
        run<22>("  same, xor waves s_setprio 3", 32, w, out, d_st, spin_s);
        run<23>("  same, no prio, xor waves 2x instrs", 48, w, out, d_st, spin_s);
        run<24>("row x16, s_setprio 2 around the bcnts", 32, w, out, d_st, spin_s);
        run<25>("two rows at once x8, prio 2 on bcnts", 32, w, out, d_st, spin_s);
        run<26>("row x16, s_setprio 2 around the XORs", 32, w, out, d_st, spin_s);
        run<27>("row+and x16, prio 3 on bcnts (17/row)", 34, w, out, d_st, spin_s);
        run<30>("row+and x16, prio 2 (17/row)", 34, w, out, d_st, spin_s);
        run<28>("  + 2 ds_read_b128 per row (kernel, R=1)", 34, w, out, d_st, spin_s);
        run<29>("  + 2 ds_read_b128 per TWO rows (R=2)", 34, w, out, d_st, spin_s);
        run<31>("v_xor with SGPR operand x256", 32, w, out, d_st, spin_s);
        run<32>("slots: even xor(SGPR), odd bcnt prio 3", 32, w, out, d_st, spin_s);
        run<33>("row x16: 8 xor(SGPR), prio 2, 8 bcnt", 32, w, out, d_st, spin_s);
        run<34>("row+and x16: xor(SGPR), prio 2 (17/row)", 34, w, out, d_st, spin_s);
        run<12>("v_bcnt x256, ONE serial chain", 32, w, out, d_st, spin_s);
        run<13>("row x16, two chains + v_add (17/row)", 34, w, out, d_st, spin_s);
        run<14>("two rows at once x8 (16 xor, 16 bcnt)", 32, w, out, d_st, spin_s);
        run<15>("row x16, alternating row buffers", 32, w, out, d_st, spin_s);
        run<16>("alternating xor,bcnt x16 rows", 32, w, out, d_st, spin_s);
    }
    return 0;
}
