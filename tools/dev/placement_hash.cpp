#include "mgx_world.h"
#include <chrono>
#include <cstdio>
#include <random>
using namespace mgx;
static uint64_t fnv(uint64_t h, const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }
int main() {
    uint64_t H = 1469598103934665603ull; double tsum = 0; long rej = 0; int calls = 0;
    for (int task = 0; task < 3; task++) {
        World w; std::string err; std::mt19937 g(task + 1);
        int nb = task == 0 ? 10 : 4;
        if (task == 1) for (int i = 0; i < 3; i++) { EntityDef q{}; q.kind = 2; q.colour = i; q.x = -0.9 + 0.6 * i; q.y = 0.8; q.h = 0.5; q.w = 0.4; q.enabled = true; w.entities.push_back(q); }
        EntityDef r{}; r.kind = 0; r.x = 0.1; r.y = -0.2; r.angle = 0.3; r.enabled = true; w.entities.push_back(r);
        for (int i = 0; i < nb; i++) { EntityDef s{}; s.kind = 1; s.shape_type = g() % 7; s.colour = g() % 4; s.x = -0.8 + 0.17 * i; s.y = 0.5 * ((i % 3) - 1); s.angle = 0.1 * i; s.enabled = true; w.entities.push_back(s); }
        if (w.finalize(100, err)) { printf("err %s\n", err.c_str()); return 1; }
        int ne = (int)w.entities.size();
        std::vector<uint8_t> en(ne, 1); std::vector<int> st(ne, -1);
        uint32_t key[624]; int pos = 624; for (auto &k : key) k = g();
        for (int k = 0; k < 1500; k++) {
            for (int i = 0; i < ne; i++) if (w.entities[i].kind == 1) { st[i] = (g() % 8) - 1; en[i] = (g() % 5) != 0; }
            World v; if (w.variant(en.data(), st.data(), v, err)) { printf("err %s\n", err.c_str()); return 1; }
            std::vector<double> poses(3 * ne), hw(2 * ne, 0.0);
            for (int i = 0; i < ne; i++) { poses[3 * i] = w.entities[i].kind == 2 ? w.entities[i].x + w.entities[i].w / 2 : w.entities[i].x; poses[3 * i + 1] = w.entities[i].kind == 2 ? w.entities[i].y - w.entities[i].h / 2 : w.entities[i].y; poses[3 * i + 2] = w.entities[i].angle; hw[2 * i] = 0.3 + 0.01 * (k % 20); hw[2 * i + 1] = 0.35; }
            std::vector<int> ents; std::vector<uint8_t> ign(ne, 0), rp, rr; std::vector<double> pl, rl;
            for (int i = 0; i < ne; i++) { if (task == 1 && w.entities[i].kind == 2 && (k & 1)) continue; ents.push_back(i); rp.push_back(1); rr.push_back(w.entities[i].kind != 2); pl.push_back((k % 3) == 0 ? 0.4 : -1.0); rl.push_back((k % 4) == 0 ? 0.5 : -1.0); }
            if (k % 7 == 0) ign[ne - 1] = 1;
            const double arena[4] = {-1, 1, -1, 1};
            auto a = std::chrono::steady_clock::now();
            int rc = v.randomise_all_poses(poses.data(), ents.data(), (int)ents.size(), ign.data(), arena, rp.data(), rr.data(), pl.data(), rl.data(), key, &pos, (k & 2) ? hw.data() : nullptr);
            tsum += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); calls++;
            if (rc > 0) rej += rc;
            H = fnv(H, &rc, 4); H = fnv(H, poses.data(), poses.size() * 8); H = fnv(H, &pos, 4);
            // single queries
            for (int i = 0; i < ne; i++) { bool c = v.placement_collides(i, poses.data(), en.data(), (k & 2) ? hw.data() : nullptr); H = fnv(H, &c, 1); }
        }
    }
    printf("%016llx  %.1f us per call, %ld rejected draws in %d calls\n", (unsigned long long)H, tsum / calls, rej, calls);
}
