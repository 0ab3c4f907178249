# generate a 4-group K step (acc[4][16]; groups 0,1 in VGPRs, 2,3 in AGPRs)
def block(cr, base):
    E, F = base, base + 8
    lines = []
    if cr == 0:
        lines += ['v_mov_b64 v[%d:%d], %%[p1]' % (E + 2, E + 3),
                  'v_mov_b64 v[%d:%d], %%[p2]' % (E + 4, E + 5),
                  'v_pk_mov_b32 v[%d:%d], %%[p0], %%[p1] op_sel:[1,0]' % (F, F + 1),
                  'v_pk_mov_b32 v[%d:%d], %%[p1], %%[p2] op_sel:[1,0]' % (F + 2, F + 3),
                  'v_pk_mov_b32 v[%d:%d], %%[p2], %%[p3] op_sel:[1,0]' % (F + 4, F + 5)]
        t0 = '%[qa]'
    else:
        for m in range(7):
            lines.append('v_alignbyte_b32 v%d, %%[w%d], %%[w%d], %d' % (E + m, m + 1, m, cr))
        for k in range(3):
            lines.append('v_pk_mov_b32 v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[1,0]' %
                         (F + 2 * k, F + 2 * k + 1, E + 2 * k, E + 2 * k + 1, E + 2 * k + 2, E + 2 * k + 3))
        t0 = 'v[%d:%d]' % (E, E + 3)
    tup = {0: t0, 2: 'v[%d:%d]' % (E + 2, E + 5), 1: 'v[%d:%d]' % (F, F + 3), 3: 'v[%d:%d]' % (F + 2, F + 5)}
    lines.append('s_nop 1')
    for cq in (0, 2, 1, 3):
        for mb in range(4):
            lines.append('v_mfma_i32_16x16x64_i8 %%[c%d%d], %%[a%d], %s, %%[c%d%d]' % (cq, mb, mb, tup[cq], cq, mb))
    text = '\\n\\t'.join(lines)
    outs = ', '.join('[c%d%d] "+%s"(acc%d[%d])' % (cq, mb, 'v' if mb < 2 else 'a', mb, 4 * cq + cr) for cq in range(4) for mb in range(4))
    ins = ', '.join('[a%d] "v"(a%d)' % (m, m) for m in range(4)) + ', '
    if cr == 0:
        ins += '[qa] "v"(qa), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2), [p3] "v"(p3)'
    else:
        ins += ', '.join('[w%d] "v"(W%d)' % (m, m) for m in range(8))
    clob = ', '.join('"v%d"' % r for r in range(base, base + 14))
    return '    asm volatile("%s"\n                 : %s\n                 : %s\n                 : %s);\n' % (text, outs, ins, clob)
code = '''__device__ __forceinline__ void step4(v4i (&acc0)[16], v4i (&acc1)[16], v4i (&acc2)[16], v4i (&acc3)[16], const v4i qa, const v4i qb,
                                      const v4i a0, const v4i a1, const v4i a2, const v4i a3) {
    typedef int v2i __attribute__((ext_vector_type(2)));
    const v2i p0 = {qa.x, qa.y}, p1 = {qa.z, qa.w}, p2 = {qb.x, qb.y}, p3 = {qb.z, qb.w};
    const int W0 = qa.x, W1 = qa.y, W2 = qa.z, W3 = qa.w, W4 = qb.x, W5 = qb.y, W6 = qb.z, W7 = qb.w;
''' + ''.join(block(cr, 228 if cr % 2 == 0 else 242) for cr in range(4)) + '}\n'
open('step4.inc','w').write(code)
