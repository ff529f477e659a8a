#include "mgx_world.h"
#include <cstdio>
#include <random>
#include <thread>
using namespace mgx;
static uint64_t fnv(uint64_t h, const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }
int main() {
    uint64_t H = 1469598103934665603ull;
    for (int task = 0; task < 3; task++) {
        World w; std::string err; std::mt19937 g(task + 1);
        int nb = task == 0 ? 10 : 4;
        if (task == 1) for (int i = 0; i < 3; i++) { EntityDef q{}; q.kind = 2; q.colour = i; q.x = -0.9 + 0.6 * i; q.y = 0.8; q.h = 0.5; q.w = 0.4; q.enabled = true; w.entities.push_back(q); }
        EntityDef r{}; r.kind = 0; r.x = 0.1; r.y = -0.2; r.angle = 0.3; r.enabled = true; w.entities.push_back(r);
        for (int i = 0; i < nb; i++) { EntityDef s{}; s.kind = 1; s.shape_type = g() % 7; s.colour = g() % 4; s.x = -0.8 + 0.17 * i; s.y = 0.5 * ((i % 3) - 1); s.angle = 0.1 * i; s.enabled = true; w.entities.push_back(s); }
        if (w.finalize(100, err)) { printf("err %s\n", err.c_str()); return 1; }
        int ne = (int)w.entities.size();
        std::vector<uint8_t> en(ne, 1); std::vector<int> st(ne, -1);
        for (int k = 0; k < 3000; k++) {
            for (int i = 0; i < ne; i++) if (w.entities[i].kind == 1) { st[i] = (g() % 8) - 1; en[i] = (g() % 5) != 0; }
            World v; if (w.variant(en.data(), st.data(), v, err)) { printf("err %s\n", err.c_str()); return 1; }
            TmplHeader h; std::vector<int32_t> iw; std::vector<double> rw, pw;
            for (int strip = 0; strip < 2; strip++) {
                v.serialise(h, iw, rw, pw, strip);
                H = fnv(H, &h, sizeof h); H = fnv(H, iw.data(), iw.size() * 4); H = fnv(H, rw.data(), rw.size() * 8); H = fnv(H, pw.data(), pw.size() * 8);
            }
        }
    }
    printf("%016llx\n", (unsigned long long)H);
}
