/* ref_kernel_harness.c -- thin C caller of the functions of the REFERENCE's own kernel file
 * (/root/reference/src/kernel/volumeraycast.cl, compiled where it lies with `clang -x cl` to x86-64;
 * oracle/Makefile target `ref`) whose call graph touches no OpenCL builtin: the hybrid Tausworthe /
 * LCG generator behind the ambient occlusion (ui_randStep, lcgStep, hybridui_rand, :48-80) and the
 * box-edge test behind showEss (checkBoundingBox, :323-343).  Every other function of that object
 * needs the OpenCL C builtin library, which this image lacks; the object is linked with those symbols
 * left undefined and loaded RTLD_LAZY -- nothing stands in for them, and nothing here calls into them.
 * TEST INFRASTRUCTURE ONLY; exists only where /root/reference exists.  Built with the same clang as
 * the kernel object so that OpenCL's vector types (ext_vector_type) are passed identically. */
typedef unsigned int uint;
typedef uint uint4 __attribute__((ext_vector_type(4)));
typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));

/* the reference's definitions (volumeraycast.cl:50, :67, :74, :323) */
extern uint ui_randStep(uint4 *ui_rand, uint p, int s1, int s2, int s3, uint m);
extern uint lcgStep(uint4 *ui_rand, uint a, uint c);
extern float hybridui_rand(uint4 *ui_rand);
extern _Bool checkBoundingBox(float3 pos, float3 voxLen, float2 bound);

uint refk_ui_rand_step(uint st[4], uint p, int s1, int s2, int s3, uint m)
{
    uint4 v = {st[0], st[1], st[2], st[3]};
    const uint r = ui_randStep(&v, p, s1, s2, s3, m);
    st[0] = v.x; st[1] = v.y; st[2] = v.z; st[3] = v.w;
    return r;
}

uint refk_lcg_step(uint st[4], uint a, uint c)
{
    uint4 v = {st[0], st[1], st[2], st[3]};
    const uint r = lcgStep(&v, a, c);
    st[0] = v.x; st[1] = v.y; st[2] = v.z; st[3] = v.w;
    return r;
}

float refk_hybrid_rand(uint st[4])
{
    uint4 v = {st[0], st[1], st[2], st[3]};
    const float r = hybridui_rand(&v);
    st[0] = v.x; st[1] = v.y; st[2] = v.z; st[3] = v.w;
    return r;
}

int refk_check_bounding_box(const float pos[3], const float voxLen[3], float b0, float b1)
{
    const float3 p = {pos[0], pos[1], pos[2]}, l = {voxLen[0], voxLen[1], voxLen[2]};
    const float2 b = {b0, b1};
    return checkBoundingBox(p, l, b) ? 1 : 0;
}
