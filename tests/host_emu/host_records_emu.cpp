// host_records_emu.cpp -- TEST INFRASTRUCTURE: the device's text / record formatters (vuke_amd/csrc/vk_lines.cuh, vk_records.cuh)
// compiled for the host, one lane per item (hip_shims.h): vk_lines_kernel, vk_verbose_kernel, vk_records_key_kernel,
// vk_records_addr_kernel write their fixed-stride slots and lengths into host memory; this program prints them, and
// tests/test_host_records.py holds them to python's own encodings (hashlib double-SHA-256 Base58Check, BIP-173 bech32, int
// formatting).  Built with -fsanitize=address,undefined: a value one byte longer than its slot is a test failure here, where on
// the GPU it would be silent corruption of the neighbouring item.  The compaction kernel shuffles across a wave and is not run.
// Build: clang++ -std=c++17 -O1 -fsanitize=address,undefined -I vuke_amd/csrc -I include -I tests/host_emu host_records_emu.cpp
#include "hip_shims.h"

#include "vuke_hip.h"
#include "vk_records.cuh"

static std::vector<unsigned char> unhex(const char *h)
{
    std::vector<unsigned char> o;
    if (h[0] == '-' && h[1] == 0) return o;
    for (size_t i = 0; h[i] && h[i + 1]; i += 2) {
        unsigned v;
        sscanf(h + i, "%2x", &v);
        o.push_back((unsigned char)v);
    }
    return o;
}
static void print_hex(const unsigned char *p, size_t n)
{
    for (size_t i = 0; i < n; i++) printf("%02x", p[i]);
}

// stdin: one item per line: "<input u64> <slot> <key 32B hex> <h160c hex> <h160u hex> <xy 64B hex> <valid 0|1>"; argv: lo, slots, the
// transform name, then one label suffix per slot ("-" = empty).  Output per item: "lines <hex>", "verbose <hex>", then one
// "col <c> <hex>" per record column (the slot's first len bytes).
int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const u64 lo = strtoull(argv[1], nullptr, 10);
    const u32 slots = (u32)atoi(argv[2]);
    if (argc != 4 + (int)slots) return 2;
    std::vector<unsigned char> texts((size_t)(slots + 1) * VK_LINE_SUFFIX_BYTES, 0);
    for (u32 i = 0; i <= slots; i++) {
        const char *src = i < slots ? argv[4 + i] : argv[3];
        if (!strcmp(src, "-")) src = "";
        texts[(size_t)i * VK_LINE_SUFFIX_BYTES] = (unsigned char)strlen(src);
        memcpy(&texts[(size_t)i * VK_LINE_SUFFIX_BYTES + 1], src, strlen(src));
    }
    std::vector<u32> priv, hc, hu, xy;
    std::vector<unsigned char> valid;
    static char line[1024], k[80], c[48], u[48], p[136];
    unsigned long long input;
    unsigned slot, v;
    std::vector<std::pair<u64, u32>> ids;
    while (fgets(line, sizeof line, stdin)) {
        if (sscanf(line, "%llu %u %79s %47s %47s %135s %u", &input, &slot, k, c, u, p, &v) != 7) continue;
        auto push = [](std::vector<u32> &dst, const std::vector<unsigned char> &b) {
            for (size_t i = 0; i < b.size(); i += 4) {
                u32 w;
                memcpy(&w, &b[i], 4);
                dst.push_back(w);
            }
        };
        push(priv, unhex(k));
        push(hc, unhex(c));
        push(hu, unhex(u));
        push(xy, unhex(p));
        valid.push_back((unsigned char)v);
        ids.push_back({input, slot});
    }
    // the kernels address items as input * slots + slot from `lo`: lay the given items out that way (holes stay invalid)
    u64 items = 0;
    for (auto &id : ids) items = std::max<u64>(items, (id.first - lo) * slots + id.second + 1);
    std::vector<u32> P(items * 8, 0), HC(items * 5, 0), HU(items * 5, 0), XY(items * 16, 0);
    std::vector<unsigned char> V(items, 0);
    for (size_t i = 0; i < ids.size(); i++) {
        const u64 at = (ids[i].first - lo) * slots + ids[i].second;
        memcpy(&P[at * 8], &priv[i * 8], 32);
        memcpy(&HC[at * 5], &hc[i * 5], 20);
        memcpy(&HU[at * 5], &hu[i * 5], 20);
        memcpy(&XY[at * 16], &xy[i * 16], 64);
        V[at] = valid[i];
    }
    // exact-size heap blocks: AddressSanitizer sees a write past the last slot; a write past an inner slot shows as a changed guard
    std::vector<u64> lslots(items * (VK_LINE_STRIDE / 8)), vslots(items * (VK_VERBOSE_STRIDE / 8));
    std::vector<u32> llen(items), vlen(items);
    vk_record_slots rs;
    std::vector<std::vector<u64>> cs(VK_RC_COLUMNS);
    std::vector<std::vector<u32>> cl(VK_RC_COLUMNS);
    for (int col = 0; col < VK_RC_COLUMNS; col++) {
        cs[col].assign(items * (vk_rc_stride(col) / 8), 0xA5A5A5A5A5A5A5A5ull);
        cl[col].assign(items, 0xFFFFFFFFu);
        rs.slot[col] = cs[col].data();
        rs.lens[col] = cl[col].data();
    }
    blockDim.x = 256;
    for (u64 it = 0; it < items; it++) {
        blockIdx.x = (unsigned)(it / 256);
        threadIdx.x = (unsigned)(it % 256);
        vk_lines_kernel(lo, items, slots, P.data(), HC.data(), V.data(), texts.data(), nullptr, lslots.data(), llen.data());
        vk_verbose_kernel(lo, items, slots, P.data(), HC.data(), HU.data(), V.data(), texts.data(), vslots.data(), vlen.data());
        vk_records_key_kernel(lo, items, slots, P.data(), V.data(), texts.data(), rs);
        vk_records_addr_kernel(items, HC.data(), HU.data(), XY.data(), V.data(), rs);
    }
    for (u64 it = 0; it < items; it++) {
        printf("item %llu %u %u\n", (unsigned long long)(lo + it / slots), (unsigned)(it % slots), V[it]);
        printf("lines ");
        print_hex((const unsigned char *)&lslots[it * (VK_LINE_STRIDE / 8)], llen[it]);
        printf("\nverbose ");
        print_hex((const unsigned char *)&vslots[it * (VK_VERBOSE_STRIDE / 8)], vlen[it]);
        printf("\n");
        for (int col = 0; col < VK_RC_COLUMNS; col++) {
            if (cl[col][it] > vk_rc_stride(col)) {
                printf("OVERRUN col %d len %u\n", col, cl[col][it]);
                return 3;
            }
            printf("col %d ", col);
            print_hex((const unsigned char *)&cs[col][it * (vk_rc_stride(col) / 8)], cl[col][it]);
            printf("\n");
        }
    }
    return 0;
}
