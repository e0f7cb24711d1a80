"""Generator of dense_hidden4_asm (bayesgm_amd/csrc/bgm_device.h): the four hidden layers of the g net as one\nhand-scheduled inline-asm block.  Prints nothing; writes /tmp/hidden4_asm.h (paste into bgm_device.h, then renumber\nthe two address operands to %8 / %9: outputs come first in the operand list)."""
P0, Q0 = 208, 224            # activation register sets v[208:223], v[224:239]
BUF = [244, 248, 252]        # A-fragment buffers
TMP = 243
NL = 4
def woff(l, s): return l * 16384 + (16 * (s >> 2) + (s & 3)) * 256
def boff(l, t): return l * 256 + t * 64
out = []
queue = []                   # outstanding LDS loads (tags), in issue order
def emit(x): out.append(x)
def load(tag, text):
    emit(text); queue.append(tag)
def wait(tag):
    if tag not in queue: return
    i = queue.index(tag)
    emit(f"s_waitcnt lgkmcnt({len(queue) - 1 - i})")
    del queue[:i + 1]
def frag_load(l, s, k):      # k = global fragment counter -> buffer
    b = BUF[k % 3]
    load(("F", l, s), f"ds_read_b128 v[{b}:{b+3}], %0 offset:{woff(l, s)}")
def lrelu(reg):                  # lrelu_s: x + (2/3)|x| = LeakyReLU(x) / 0.6, v{TMP} holds 2/3
    emit(f"v_fma_f32 v{reg}, |v{reg}|, v{TMP}, v{reg}")
emit("s_waitcnt lgkmcnt(0)")
emit(f"v_mov_b32 v{TMP}, 0x3f2aaaab")
for t in range(4):
    load(("B", 0, t), f"ds_read_b128 v[{Q0+4*t}:{Q0+4*t+3}], %1 offset:{boff(0, t)}")
frag_load(0, 0, 0); frag_load(0, 1, 1)
k = 0                        # global step counter
for l in range(NL):
    I, O = (P0, Q0) if l % 2 == 0 else (Q0, P0)
    for s in range(16):
        wait(("F", l, s))
        if s == 0:
            for t in range(3): wait(("B", l, t))
        if l > 0 and s + 1 <= 15: lrelu(I + s + 1)          # just in time for the next step
        b = BUF[k % 3]
        emit(f"v_mfma_f32_16x16x4_f32 v[{O}:{O+3}], v{b}, v{I+s}, v[{O}:{O+3}]")
        # request the fragment two steps ahead (crossing into the next layer)
        s2, l2 = s + 2, l
        if s2 > 15: s2, l2 = s2 - 16, l + 1
        if l2 < NL: frag_load(l2, s2, k + 2)
        for t in range(1, 4):
            if s == 0 and t == 3: wait(("B", l, 3))
            emit(f"v_mfma_f32_16x16x4_f32 v[{O+4*t}:{O+4*t+3}], v{b+t}, v{I+s}, v[{O+4*t}:{O+4*t+3}]")
        if (s & 3) == 3 and l + 1 < NL:                      # input tuple s>>2 is dead: it becomes next layer's accumulator
            t = s >> 2
            load(("B", l + 1, t), f"ds_read_b128 v[{I+4*t}:{I+4*t+3}], %1 offset:{boff(l + 1, t)}")
        k += 1
    if l + 1 < NL:
        emit("s_nop 15")                                     # MFMA -> VALU read of the new input set (tuple 0 first)
        lrelu(O + 0)
emit("s_nop 15")
assert not queue or all(q[0] != "F" for q in queue), queue
emit("s_waitcnt lgkmcnt(0)")
body = "\n".join('      "%s\\n"' % x for x in out)
final_set = "P" if (NL - 1) % 2 == 1 else "Q"
code = '''
// ---------------------------------------------------------------------------------------------
// The four hidden->hidden layers of the g net (64 x 64 each) as ONE hand-scheduled block.
// Between two layers the compiler-scheduled path pays bias loads + the first A fragments + the MFMA drain + a
// 32-instruction LeakyReLU, ~580 cycles per layer with the matrix pipe empty (measured).  Here the activation sets
// live in fixed registers (P = v208-v223, Q = v224-v239) and alternate as B operands / accumulators; a layer's
// accumulator tuples are loaded with the bias as soon as the previous layer has consumed them as inputs, the A
// fragments stream two K-steps ahead ACROSS layer boundaries, and LeakyReLU is applied to one input element per
// K-step, just in time.  Software-managed hazards: counted lgkmcnt per LDS load, s_nop between the last MFMA of a
// layer and the first VALU read of its result.
//   w_addr: LDS byte address of this lane's fragment of layer 0, K-step 0 (layers are 16 KiB apart);
//   b_addr: LDS byte address of bias feature 4g of layer 0 (layers 256 B apart).
//   p: in = activated input, out = RAW output of the 4th layer (caller applies LeakyReLU);  q: scratch set.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void dense_hidden4_asm(unsigned w_addr, unsigned b_addr, f32x4 (&p)[4], f32x4 (&q)[4]) {
  asm volatile(
''' + body + '''
      : "+{v[208:211]}"(p[0]), "+{v[212:215]}"(p[1]), "+{v[216:219]}"(p[2]), "+{v[220:223]}"(p[3]),
        "=&{v[224:227]}"(q[0]), "=&{v[228:231]}"(q[1]), "=&{v[232:235]}"(q[2]), "=&{v[236:239]}"(q[3])
      : "v"(w_addr), "v"(b_addr)
      : "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255");
}
'''
assert final_set == "P"
open("/tmp/hidden4_asm.h", "w").write(code)
print(len(out), "asm lines")
