// Batched LDS reads for gfx950, issued from one asm block each (internal to the IK kernels).
//
// Why asm: whatever the scheduling strategy (default, max-ilp) and whatever the register budget, hipcc's
// machine scheduler places every ds_read directly in front of its first use when the consumers form a
// dependent chain, i.e. one exposed LDS latency (~100 cycles at low occupancy) per read.  In the Riccati
// kernel that was half of the run time.  Each helper issues its reads back to back and waits once;
// outputs are early-clobber so they cannot alias the address register, "memory" orders the block after
// the preceding LDS writes.  Addresses are LDS byte offsets (lds_offset()).
#pragma once
#include <hip/hip_runtime.h>

namespace bunmpc {

typedef double double2_t __attribute__((ext_vector_type(2)));

// byte offset of a __shared__ object inside the workgroup's LDS (low half of its flat address)
__device__ __forceinline__ unsigned lds_offset(const void *p) { return (unsigned)(unsigned long long)p; }

// 3 x 16 bytes from a 16-byte aligned address
__device__ __forceinline__ void lds_read_b128x3(unsigned addr, double2_t (&o)[3]) {
    asm volatile("ds_read_b128 %0, %3 offset:0\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2])
                 : "v"(addr) : "memory");
}
// 8 x 16 bytes (16 doubles) from a 16-byte aligned address
__device__ __forceinline__ void lds_read_b128x8(unsigned addr, double2_t (&o)[8]) {
    asm volatile("ds_read_b128 %0, %8 offset:0\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\tds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
                 : "v"(addr) : "memory");
}
// 9 x 16 bytes (18 doubles) from a 16-byte aligned address
__device__ __forceinline__ void lds_read_b128x9(unsigned addr, double2_t (&o)[9]) {
    asm volatile("ds_read_b128 %0, %9 offset:0\n\t" "ds_read_b128 %1, %9 offset:16\n\t" "ds_read_b128 %2, %9 offset:32\n\t" "ds_read_b128 %3, %9 offset:48\n\t" "ds_read_b128 %4, %9 offset:64\n\t" "ds_read_b128 %5, %9 offset:80\n\t" "ds_read_b128 %6, %9 offset:96\n\t" "ds_read_b128 %7, %9 offset:112\n\t" "ds_read_b128 %8, %9 offset:128\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8])
                 : "v"(addr) : "memory");
}
// 11 x 16 bytes (22 doubles) from a 16-byte aligned address
__device__ __forceinline__ void lds_read_b128x11(unsigned addr, double2_t (&o)[11]) {
    asm volatile("ds_read_b128 %0, %11 offset:0\n\tds_read_b128 %1, %11 offset:16\n\tds_read_b128 %2, %11 offset:32\n\tds_read_b128 %3, %11 offset:48\n\tds_read_b128 %4, %11 offset:64\n\tds_read_b128 %5, %11 offset:80\n\tds_read_b128 %6, %11 offset:96\n\tds_read_b128 %7, %11 offset:112\n\tds_read_b128 %8, %11 offset:128\n\tds_read_b128 %9, %11 offset:144\n\tds_read_b128 %10, %11 offset:160\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10])
                 : "v"(addr) : "memory");
}
// the first ten doubles of five records of 22 doubles (the line search's part sums: mass, first moment, momentum): [5 pa + j] = doubles 2j, 2j + 1 of record pa
__device__ __forceinline__ void lds_read_parts_head(unsigned addr, double2_t (&o)[25]) {
    asm volatile("ds_read_b128 %0, %25 offset:0\n\tds_read_b128 %1, %25 offset:16\n\tds_read_b128 %2, %25 offset:32\n\tds_read_b128 %3, %25 offset:48\n\tds_read_b128 %4, %25 offset:64\n\tds_read_b128 %5, %25 offset:176\n\tds_read_b128 %6, %25 offset:192\n\tds_read_b128 %7, %25 offset:208\n\tds_read_b128 %8, %25 offset:224\n\tds_read_b128 %9, %25 offset:240\n\tds_read_b128 %10, %25 offset:352\n\tds_read_b128 %11, %25 offset:368\n\tds_read_b128 %12, %25 offset:384\n\tds_read_b128 %13, %25 offset:400\n\tds_read_b128 %14, %25 offset:416\n\tds_read_b128 %15, %25 offset:528\n\tds_read_b128 %16, %25 offset:544\n\tds_read_b128 %17, %25 offset:560\n\tds_read_b128 %18, %25 offset:576\n\tds_read_b128 %19, %25 offset:592\n\tds_read_b128 %20, %25 offset:704\n\tds_read_b128 %21, %25 offset:720\n\tds_read_b128 %22, %25 offset:736\n\tds_read_b128 %23, %25 offset:752\n\tds_read_b128 %24, %25 offset:768\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]), "=&v"(o[16]), "=&v"(o[17]), "=&v"(o[18]), "=&v"(o[19]), "=&v"(o[20]), "=&v"(o[21]), "=&v"(o[22]), "=&v"(o[23]), "=&v"(o[24])
                 : "v"(addr) : "memory");
}
// three consecutive doubles of each of five records of 22 doubles, from the address of the first record's triple: [3 pa + c]
__device__ __forceinline__ void lds_read_parts_triple(unsigned addr, double (&o)[15]) {
    asm volatile("ds_read_b64 %0, %15 offset:0\n\tds_read_b64 %1, %15 offset:8\n\tds_read_b64 %2, %15 offset:16\n\tds_read_b64 %3, %15 offset:176\n\tds_read_b64 %4, %15 offset:184\n\tds_read_b64 %5, %15 offset:192\n\tds_read_b64 %6, %15 offset:352\n\tds_read_b64 %7, %15 offset:360\n\tds_read_b64 %8, %15 offset:368\n\tds_read_b64 %9, %15 offset:528\n\tds_read_b64 %10, %15 offset:536\n\tds_read_b64 %11, %15 offset:544\n\tds_read_b64 %12, %15 offset:704\n\tds_read_b64 %13, %15 offset:712\n\tds_read_b64 %14, %15 offset:720\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14])
                 : "v"(addr) : "memory");
}
// 18 x 16 bytes (36 doubles) from a 16-byte aligned address
__device__ __forceinline__ void lds_read_b128x18(unsigned addr, double2_t (&o)[18]) {
    asm volatile("ds_read_b128 %0, %18 offset:0\n\tds_read_b128 %1, %18 offset:16\n\tds_read_b128 %2, %18 offset:32\n\tds_read_b128 %3, %18 offset:48\n\tds_read_b128 %4, %18 offset:64\n\tds_read_b128 %5, %18 offset:80\n\tds_read_b128 %6, %18 offset:96\n\tds_read_b128 %7, %18 offset:112\n\tds_read_b128 %8, %18 offset:128\n\tds_read_b128 %9, %18 offset:144\n\tds_read_b128 %10, %18 offset:160\n\tds_read_b128 %11, %18 offset:176\n\tds_read_b128 %12, %18 offset:192\n\tds_read_b128 %13, %18 offset:208\n\tds_read_b128 %14, %18 offset:224\n\tds_read_b128 %15, %18 offset:240\n\tds_read_b128 %16, %18 offset:256\n\tds_read_b128 %17, %18 offset:272\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]), "=&v"(o[16]), "=&v"(o[17])
                 : "v"(addr) : "memory");
}
// 22 x 16 bytes (44 doubles) from a 16-byte aligned address
__device__ __forceinline__ void lds_read_b128x22(unsigned addr, double2_t (&o)[22]) {
    asm volatile("ds_read_b128 %0, %22 offset:0\n\tds_read_b128 %1, %22 offset:16\n\tds_read_b128 %2, %22 offset:32\n\tds_read_b128 %3, %22 offset:48\n\tds_read_b128 %4, %22 offset:64\n\tds_read_b128 %5, %22 offset:80\n\tds_read_b128 %6, %22 offset:96\n\tds_read_b128 %7, %22 offset:112\n\tds_read_b128 %8, %22 offset:128\n\tds_read_b128 %9, %22 offset:144\n\tds_read_b128 %10, %22 offset:160\n\tds_read_b128 %11, %22 offset:176\n\tds_read_b128 %12, %22 offset:192\n\tds_read_b128 %13, %22 offset:208\n\tds_read_b128 %14, %22 offset:224\n\tds_read_b128 %15, %22 offset:240\n\tds_read_b128 %16, %22 offset:256\n\tds_read_b128 %17, %22 offset:272\n\tds_read_b128 %18, %22 offset:288\n\tds_read_b128 %19, %22 offset:304\n\tds_read_b128 %20, %22 offset:320\n\tds_read_b128 %21, %22 offset:336\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]), "=&v"(o[16]), "=&v"(o[17]), "=&v"(o[18]), "=&v"(o[19]), "=&v"(o[20]), "=&v"(o[21])
                 : "v"(addr) : "memory");
}
// 20 x 16 bytes (40 doubles) from a 16-byte aligned address
__device__ __forceinline__ void lds_read_b128x20(unsigned addr, double2_t (&o)[20]) {
    asm volatile("ds_read_b128 %0, %20 offset:0\n\tds_read_b128 %1, %20 offset:16\n\tds_read_b128 %2, %20 offset:32\n\tds_read_b128 %3, %20 offset:48\n\tds_read_b128 %4, %20 offset:64\n\tds_read_b128 %5, %20 offset:80\n\tds_read_b128 %6, %20 offset:96\n\tds_read_b128 %7, %20 offset:112\n\tds_read_b128 %8, %20 offset:128\n\tds_read_b128 %9, %20 offset:144\n\tds_read_b128 %10, %20 offset:160\n\tds_read_b128 %11, %20 offset:176\n\tds_read_b128 %12, %20 offset:192\n\tds_read_b128 %13, %20 offset:208\n\tds_read_b128 %14, %20 offset:224\n\tds_read_b128 %15, %20 offset:240\n\tds_read_b128 %16, %20 offset:256\n\tds_read_b128 %17, %20 offset:272\n\tds_read_b128 %18, %20 offset:288\n\tds_read_b128 %19, %20 offset:304\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14]), "=&v"(o[15]), "=&v"(o[16]), "=&v"(o[17]), "=&v"(o[18]), "=&v"(o[19])
                 : "v"(addr) : "memory");
}
// 18 x 16 bytes from a 16-byte aligned address plus four consecutive doubles from an 8-byte aligned one, one wait
__device__ __forceinline__ void lds_read_b128x18_and4(unsigned kaddr, unsigned waddr, double2_t (&k)[18], double2_t (&w)[2]) {
    asm volatile("ds_read_b128 %0, %20 offset:0\n\tds_read_b128 %1, %20 offset:16\n\tds_read_b128 %2, %20 offset:32\n\tds_read_b128 %3, %20 offset:48\n\tds_read_b128 %4, %20 offset:64\n\tds_read_b128 %5, %20 offset:80\n\tds_read_b128 %6, %20 offset:96\n\tds_read_b128 %7, %20 offset:112\n\tds_read_b128 %8, %20 offset:128\n\tds_read_b128 %9, %20 offset:144\n\tds_read_b128 %10, %20 offset:160\n\tds_read_b128 %11, %20 offset:176\n\tds_read_b128 %12, %20 offset:192\n\tds_read_b128 %13, %20 offset:208\n\tds_read_b128 %14, %20 offset:224\n\tds_read_b128 %15, %20 offset:240\n\tds_read_b128 %16, %20 offset:256\n\tds_read_b128 %17, %20 offset:272\n\tds_read2_b64 %18, %21 offset0:0 offset1:1\n\tds_read2_b64 %19, %21 offset0:2 offset1:3\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(k[0]), "=&v"(k[1]), "=&v"(k[2]), "=&v"(k[3]), "=&v"(k[4]), "=&v"(k[5]), "=&v"(k[6]), "=&v"(k[7]), "=&v"(k[8]), "=&v"(k[9]), "=&v"(k[10]), "=&v"(k[11]), "=&v"(k[12]), "=&v"(k[13]), "=&v"(k[14]), "=&v"(k[15]), "=&v"(k[16]), "=&v"(k[17]), "=&v"(w[0]), "=&v"(w[1])
                 : "v"(kaddr), "v"(waddr) : "memory");
}
// 36 consecutive doubles (8-byte aligned) with the 18 ds_read2_b64 in flight together
__device__ __forceinline__ void lds_read_row36(unsigned addr, double (&x)[36]) {
    double2_t t[18];
    asm volatile("ds_read2_b64 %0, %18 offset0:0 offset1:1\n\tds_read2_b64 %1, %18 offset0:2 offset1:3\n\tds_read2_b64 %2, %18 offset0:4 offset1:5\n\tds_read2_b64 %3, %18 offset0:6 offset1:7\n\tds_read2_b64 %4, %18 offset0:8 offset1:9\n\tds_read2_b64 %5, %18 offset0:10 offset1:11\n\tds_read2_b64 %6, %18 offset0:12 offset1:13\n\tds_read2_b64 %7, %18 offset0:14 offset1:15\n\tds_read2_b64 %8, %18 offset0:16 offset1:17\n\tds_read2_b64 %9, %18 offset0:18 offset1:19\n\tds_read2_b64 %10, %18 offset0:20 offset1:21\n\tds_read2_b64 %11, %18 offset0:22 offset1:23\n\tds_read2_b64 %12, %18 offset0:24 offset1:25\n\tds_read2_b64 %13, %18 offset0:26 offset1:27\n\tds_read2_b64 %14, %18 offset0:28 offset1:29\n\tds_read2_b64 %15, %18 offset0:30 offset1:31\n\tds_read2_b64 %16, %18 offset0:32 offset1:33\n\tds_read2_b64 %17, %18 offset0:34 offset1:35\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7]), "=&v"(t[8]), "=&v"(t[9]), "=&v"(t[10]), "=&v"(t[11]), "=&v"(t[12]), "=&v"(t[13]), "=&v"(t[14]), "=&v"(t[15]), "=&v"(t[16]), "=&v"(t[17])
                 : "v"(addr) : "memory");
    _Pragma("unroll") for (int i = 0; i < 18; ++i) { x[2 * i] = t[i].x; x[2 * i + 1] = t[i].y; }
}
// 36 consecutive doubles of a row of its own per lane, as 36 single ds_read_b64 (two batches of 18).  Against lds_read_row36's
// ds_read2_b64: that instruction is banked over 32 banks in groups of 16 lanes, so rows r and r + 16 collide whatever the (odd)
// leading dimension -- and the Riccati wave's lanes 36..53 shadow rows 18..35 beside lanes 32..35's rows 32..35 (4 conflict cycles
// per instruction, tools/lds_conflict_probe.hip); ds_read_b64 is banked over 64 banks in groups of 32 lanes, where those rows are
// distinct (the shadow of a row in the same group reads the same address: a broadcast).
__device__ __forceinline__ void lds_read_row36_b64(unsigned addr, double (&x)[36]) {
    asm volatile("ds_read_b64 %0, %18 offset:0\n\t" "ds_read_b64 %1, %18 offset:8\n\t" "ds_read_b64 %2, %18 offset:16\n\t" "ds_read_b64 %3, %18 offset:24\n\t" "ds_read_b64 %4, %18 offset:32\n\t" "ds_read_b64 %5, %18 offset:40\n\t" "ds_read_b64 %6, %18 offset:48\n\t" "ds_read_b64 %7, %18 offset:56\n\t" "ds_read_b64 %8, %18 offset:64\n\t" "ds_read_b64 %9, %18 offset:72\n\t" "ds_read_b64 %10, %18 offset:80\n\t" "ds_read_b64 %11, %18 offset:88\n\t" "ds_read_b64 %12, %18 offset:96\n\t" "ds_read_b64 %13, %18 offset:104\n\t" "ds_read_b64 %14, %18 offset:112\n\t" "ds_read_b64 %15, %18 offset:120\n\t" "ds_read_b64 %16, %18 offset:128\n\t" "ds_read_b64 %17, %18 offset:136\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9]), "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13]), "=&v"(x[14]), "=&v"(x[15]), "=&v"(x[16]), "=&v"(x[17])
                 : "v"(addr) : "memory");
    asm volatile("ds_read_b64 %0, %18 offset:144\n\t" "ds_read_b64 %1, %18 offset:152\n\t" "ds_read_b64 %2, %18 offset:160\n\t" "ds_read_b64 %3, %18 offset:168\n\t" "ds_read_b64 %4, %18 offset:176\n\t" "ds_read_b64 %5, %18 offset:184\n\t" "ds_read_b64 %6, %18 offset:192\n\t" "ds_read_b64 %7, %18 offset:200\n\t" "ds_read_b64 %8, %18 offset:208\n\t" "ds_read_b64 %9, %18 offset:216\n\t" "ds_read_b64 %10, %18 offset:224\n\t" "ds_read_b64 %11, %18 offset:232\n\t" "ds_read_b64 %12, %18 offset:240\n\t" "ds_read_b64 %13, %18 offset:248\n\t" "ds_read_b64 %14, %18 offset:256\n\t" "ds_read_b64 %15, %18 offset:264\n\t" "ds_read_b64 %16, %18 offset:272\n\t" "ds_read_b64 %17, %18 offset:280\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[18]), "=&v"(x[19]), "=&v"(x[20]), "=&v"(x[21]), "=&v"(x[22]), "=&v"(x[23]), "=&v"(x[24]), "=&v"(x[25]), "=&v"(x[26]), "=&v"(x[27]), "=&v"(x[28]), "=&v"(x[29]), "=&v"(x[30]), "=&v"(x[31]), "=&v"(x[32]), "=&v"(x[33]), "=&v"(x[34]), "=&v"(x[35])
                 : "v"(addr) : "memory");
}
// 36 doubles at a stride of 37 doubles (a column of a row-major 36 x 37 staging matrix): two batches of 18 ds_read_b64
__device__ __forceinline__ void lds_read_col36_ld37(unsigned addr, double (&x)[36]) {
    asm volatile("ds_read_b64 %0, %18 offset:0\n\tds_read_b64 %1, %18 offset:296\n\tds_read_b64 %2, %18 offset:592\n\tds_read_b64 %3, %18 offset:888\n\tds_read_b64 %4, %18 offset:1184\n\tds_read_b64 %5, %18 offset:1480\n\tds_read_b64 %6, %18 offset:1776\n\tds_read_b64 %7, %18 offset:2072\n\tds_read_b64 %8, %18 offset:2368\n\tds_read_b64 %9, %18 offset:2664\n\tds_read_b64 %10, %18 offset:2960\n\tds_read_b64 %11, %18 offset:3256\n\tds_read_b64 %12, %18 offset:3552\n\tds_read_b64 %13, %18 offset:3848\n\tds_read_b64 %14, %18 offset:4144\n\tds_read_b64 %15, %18 offset:4440\n\tds_read_b64 %16, %18 offset:4736\n\tds_read_b64 %17, %18 offset:5032\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9]), "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13]), "=&v"(x[14]), "=&v"(x[15]), "=&v"(x[16]), "=&v"(x[17])
                 : "v"(addr) : "memory");
    asm volatile("ds_read_b64 %0, %18 offset:5328\n\tds_read_b64 %1, %18 offset:5624\n\tds_read_b64 %2, %18 offset:5920\n\tds_read_b64 %3, %18 offset:6216\n\tds_read_b64 %4, %18 offset:6512\n\tds_read_b64 %5, %18 offset:6808\n\tds_read_b64 %6, %18 offset:7104\n\tds_read_b64 %7, %18 offset:7400\n\tds_read_b64 %8, %18 offset:7696\n\tds_read_b64 %9, %18 offset:7992\n\tds_read_b64 %10, %18 offset:8288\n\tds_read_b64 %11, %18 offset:8584\n\tds_read_b64 %12, %18 offset:8880\n\tds_read_b64 %13, %18 offset:9176\n\tds_read_b64 %14, %18 offset:9472\n\tds_read_b64 %15, %18 offset:9768\n\tds_read_b64 %16, %18 offset:10064\n\tds_read_b64 %17, %18 offset:10360\n\t" "s_waitcnt lgkmcnt(0)"
                 : "=&v"(x[18]), "=&v"(x[19]), "=&v"(x[20]), "=&v"(x[21]), "=&v"(x[22]), "=&v"(x[23]), "=&v"(x[24]), "=&v"(x[25]), "=&v"(x[26]), "=&v"(x[27]), "=&v"(x[28]), "=&v"(x[29]), "=&v"(x[30]), "=&v"(x[31]), "=&v"(x[32]), "=&v"(x[33]), "=&v"(x[34]), "=&v"(x[35])
                 : "v"(addr) : "memory");
}

}  // namespace bunmpc
