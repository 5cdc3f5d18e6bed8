// jf_packed.h -- complex float32 arithmetic on (re, im) register pairs with the packed f32 VALU
// instructions of gfx950 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two float32 operations
// per lane and issue slot, which is where the f32 vector peak comes from).
//
// One packed instruction does a complex add, and a complex multiply is a packed multiply plus a packed
// FMA.  Multiplications by +-i, conjugations and real/imaginary broadcasts cost nothing: they are the
// op_sel / op_sel_hi (which half of a source feeds the low / high result) and neg_lo / neg_hi
// modifiers of the VOP3P encoding.  Measured issue cost per wave64 instruction on MI355X
// (profiles/micro/pk_rate.hip): v_pk_* about 5 cycles whatever the modifiers, v_fma_f32 4.3,
// v_add/v_mul/v_mov_b32 2.7 -- so packing pays for multiply-accumulate work (the reverb's partition
// MAC, jf_reverb.hip: 1.3x) and not for add-dominated butterflies (the spatialiser's FFTs: a fully
// packed build of jf_kernels.hip, profiles/r01_packed_kernel_experiment.patch, issues 34 % fewer VALU
// instructions and runs in the same time).  The compiler does not form these from scalar code (its SLP
// vectoriser pairs unrelated values and pays for it in v_mov shuffles), so the operations that need
// modifiers are written as one-instruction asm; plain adds, subtracts and FMAs on pairs are left
// to the compiler, which emits the packed forms for ext_vector_type(2) operands.
#pragma once
#include <hip/hip_runtime.h>

namespace jf {

#ifndef JF_DEV
#define JF_DEV __device__ __forceinline__
#endif

typedef float c2 __attribute__((ext_vector_type(2)));  // .x = re (low register), .y = im (high register)

JF_DEV c2 mk2(float re, float im) { return c2{re, im}; }
JF_DEV c2 c2_of(float2 v) { return c2{v.x, v.y}; }
JF_DEV float2 f2_of(c2 v) { return make_float2(v.x, v.y); }

#define JF_PK2(name, text)                            \
    JF_DEV c2 name(c2 a, c2 b) {                      \
        c2 r;                                         \
        asm(text : "=v"(r) : "v"(a), "v"(b));         \
        return r;                                     \
    }
// acc + (selected halves of a) * (selected halves of b).  Three-address form (the result is its own operand, not tied
// to the addend): with the addend tied ("+v") every accumulator that is carried around a loop through a branchy body was
// copied once per use -- 16 v_mov_b64 per half-filter of the pair kernel -- because the register allocator could not put
// the loop's incoming and outgoing value into one register; a free destination lets it write where the value is wanted.
#define JF_PK3(name, text)                                          \
    JF_DEV c2 name(c2 a, c2 b, c2 acc) {                            \
        c2 r;                                                       \
        asm(text : "=v"(r) : "v"(a), "v"(b), "v"(acc));             \
        return r;                                                   \
    }

// ---- additions with a rotated or conjugated second operand
JF_PK2(padd_i, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")   // a + i b
JF_PK2(psub_i, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")   // a - i b
JF_PK2(padd_c, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")                                // a + conj b
JF_PK2(psub_c, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]")                                // a - conj b
JF_PK2(pcadd_ic, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]") // conj a + i conj b

// ---- products with one half of an operand broadcast
JF_PK2(pmul_re, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]")                 // (a.re b.re, a.re b.im)
JF_PK2(pmul_blo, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]")                // a * b.lo (both halves)
JF_PK2(pmul_bhi, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]")                   // a * b.hi (both halves)
JF_PK3(pfma_blo, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]")          // acc + a * b.lo
JF_PK3(pfma_bhi, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]")             // acc + a * b.hi

// ---- a pair times / plus-times a wave-uniform scalar held as a scalar-register pair (w, w)
JF_DEV c2 pmul_s(c2 a, c2 w) {
    c2 r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(w));
    return r;
}
JF_DEV c2 pfma_s(c2 a, c2 w, c2 acc) {
    c2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(w), "v"(acc));
    return r;
}

// ---- complex products: first a packed multiply by a.re, then a packed FMA by a.im
// a * w
JF_DEV c2 pcmul(c2 a, c2 w) {
    c2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(w));                             // (a.re w.re, a.re w.im)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(r) : "v"(a), "v"(w));  // + (-a.im w.im, a.im w.re)
    return r;
}
// acc + a * w
JF_DEV c2 pcmac(c2 a, c2 w, c2 acc) {
    c2 t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(w), "v"(acc));                               // + (a.re w.re, a.re w.im)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));  // + (-a.im w.im, a.im w.re)
    return r;
}
// acc + (a.lo w.lo, a.hi w.hi) and acc + a.lo (-w.hi, w.lo): the two halves of pcmac with the first operand's halves
// given separately -- a complex product when a = (x.re, x.re) then (x.im, .), an element-by-element product when
// a = (x.re, x.im) then (0, .) (two real bins that travel as one complex entry)
JF_PK3(pfma_each, "v_pk_fma_f32 %0, %1, %2, %3")
JF_PK3(pfma_lo_rot, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]")
// acc + (a.hi, -a.lo) * b.hi  (a rotated by -90 degrees, times a real factor kept in the high half of b)
JF_PK3(pfma_nrot_bhi, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]")
// (a.lo b.lo, -a.hi b.lo)
JF_PK2(pmul_blo_conj, "v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] neg_hi:[0,1]")
// a * conj(w)
JF_DEV c2 pcmulc(c2 a, c2 w) {
    c2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(w));                // (a.re w.re, -a.re w.im)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "+v"(r) : "v"(a), "v"(w));        // + (a.im w.im, a.im w.re)
    return r;
}
// -(a * w)
JF_DEV c2 pcmul_neg(c2 a, c2 w) {
    c2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(w));   // (-a.re w.re, -a.re w.im)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "+v"(r) : "v"(a), "v"(w));  // + (a.im w.im, -a.im w.re)
    return r;
}
// a * (w.im + i w.re): the operand's halves swapped (e.g. (sin, cos) from a (cos, sin) pair)
JF_DEV c2 pcmul_sw(c2 a, c2 w) {
    c2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(a), "v"(w));                // (a.re w.im, a.re w.re)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0]" : "+v"(r) : "v"(a), "v"(w));  // + (-a.im w.re, a.im w.im)
    return r;
}
// a * (-w.re + i w.im)
JF_DEV c2 pcmul_nre(c2 a, c2 w) {
    c2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(w));                // (-a.re w.re, a.re w.im)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "+v"(r) : "v"(a), "v"(w));  // + (-a.im w.im, -a.im w.re)
    return r;
}

#undef JF_PK2
#undef JF_PK3

}  // namespace jf
