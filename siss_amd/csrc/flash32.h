// Interface between flash_attn.hip's launchers and flash_attn32.hip's 32x32x16-MFMA kernels (narrow heads; round 6).
#pragma once
struct FA32Args {
    const void *q, *k, *v, *o, *d_o;
    void *dq, *dk, *dv;
    long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;      // row strides in elements (merged layout: head h at column h * D)
    const float* lse2;                                      // [Bf * H][Sq] base-2 log-sum-exp of the forward
    float* delta;                                           // [nB * H][Sq] scratch: rowsum(dO o O), written by the dQ kernel
    int nB, Bf, H, D, Sq, Sk;                               // cotangent batch entries against Bf forward ones (entry b uses b % Bf)
    float scale;
    int pre;                                                // q holds scale * log2(e) * Q
};
bool siss_fa32_bwd_takes(const FA32Args& a);
int siss_fa32_bwd(const FA32Args& a, void* stream);
struct FA32FwdArgs {
    const void *q, *k, *v;
    void* o;
    long ldq, ldk, ldv, ldo;
    float* lse2;                                            // [B * H][Sq]
    int B, H, D, Sq, Sk;
    float scale;
    int pre;
};
bool siss_fa32_fwd_takes(const FA32FwdArgs& a);
int siss_fa32_fwd(const FA32FwdArgs& a, void* stream);
