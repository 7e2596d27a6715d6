// host_field_emu.cpp -- TEST INFRASTRUCTURE: the generated field layer (vuke_amd/csrc/vk_field.cuh) compiled for the host with its
// two macro-named headers pointed at this directory (every inline-asm statement replaced by calls into a one-lane instruction
// emulator: vk_field_prims_host.h, vk_field_gen_host.h) and
// driven from stdin: lines "op a_hex64 b_hex64" -> one line of 64 hex digits per input (canonical result).  The op numbers are
// those of vk_debug_field / vk_field_kernel (vk_aux_kernels.cuh).  Built and run by tests/test_host_field.py.
#define VK_FIELD_PRIMS_H "vk_field_prims_host.h"
#define VK_FIELD_GEN_H "vk_field_gen_host.h"
#include "vk_field.cuh"

#include <cstdio>
#include <cstring>

static void from_hex(fe &x, const char *h)
{
    for (int i = 0; i < 8; i++) {
        u32 v = 0;
        for (int j = 0; j < 8; j++) {
            const char c = h[8 * (7 - i) + j];
            v = (v << 4) | (u32)(c <= '9' ? c - '0' : (c | 32) - 'a' + 10);
        }
        x.v[i] = v;
    }
}

int main()
{
    char line[256], ha[80], hb[80];
    int op;
    while (fgets(line, sizeof line, stdin)) {
        if (sscanf(line, "%d %64s %64s", &op, ha, hb) != 3) continue;
        fe x, y, r;
        from_hex(x, ha);
        from_hex(y, hb);
        switch (op) { // the same cases as vk_field_kernel
        case 0: fe_mul(r, x, y); break;
        case 1: fe_sqr(r, x); break;
        case 2: fe_inv(r, x); break;
        case 3: fe_add(r, x, y); break;
        case 5: fe_sub3(r, x, y, y); break;
        case 6: r = x; fe_sub3(r, r, y, r); break;
        case 7: fe_add(r, x, y); fe_sub3(r, x, y, r); break;
        case 8: r = x; fe_mul(r, r, y); break;
        case 9: r = y; fe_mul(r, x, r); break;
        case 10: r = x; fe_mul(r, r, r); break;
        case 11: r = x; fe_sqr(r, r); break;
        case 12: r = x; fe_add(r, r, r); break;
        case 13: r = x; fe_sub(r, y, r); break;
        case 14: r = x; fe_sub3(r, r, r, r); break;
        case 15: r = x; fe_sqr_n(r, r, 5); fe_mul(r, r, y); break;
        case 16: fe_mul2(r, x, y, y, x); break;                        // 2xy: the 17-limb sum at its largest for x = y = 2^256 - 1
        case 17: fe_mul2(r, x, x, y, y); break;                        // x^2 + y^2
        case 18: r = x; fe_mul2(r, r, y, y, r); break;                 // result aliasing the first and the last operand
        case 19: r = y; fe_sqr(r, r); fe_mul2(r, x, r, r, y); break;   // x y^2 + y^3: result aliasing the inner operands
        case 102: r = fe_mul2_exact(x, y, y, x); break;
        case 100: r = fe_mul_exact(x, y); break; // the exact (three-chain) path on its own
        case 101: r = fe_sqr_exact(x); break;
        default: fe_sub(r, x, y); break;
        }
        fe_canon(r);
        for (int i = 7; i >= 0; i--) printf("%08x", r.v[i]);
        printf("\n");
    }
    return 0;
}
