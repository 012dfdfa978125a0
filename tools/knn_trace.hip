// Experiment: per-workgroup timeline of the kNN kernel (start/end stamps + placement).  Not part of the product.
#define DMET_KNN_STAMP 1
#include "../deepmetv2_amd/csrc/knn.hip"
#include "../deepmetv2_amd/csrc/misc.hip"
#include <vector>
#include <algorithm>
#include <map>
int main(int argc, char** argv) {
    int B = argc > 1 ? atoi(argv[1]) : 16; const int n = 4500, D = 32, k = 16;
    int64_t N = (int64_t)B * n;
    std::vector<float> hx(N * D); unsigned s = 12345; for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 8388608.f - 1.f; }
    std::vector<int64_t> hp(B + 1); for (int b = 0; b <= B; ++b) hp[b] = (int64_t)b * n;
    float *x, *dist; int64_t* ptr; int32_t* nbr; void* ws;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&ptr, hp.size() * 8); hipMalloc(&nbr, N * k * 4); hipMalloc(&dist, N * k * 4);
    size_t wsb = dmet_knn_workspace_bytes(N, B, D, k); hipMalloc(&ws, wsb);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ptr, hp.data(), hp.size() * 8, hipMemcpyHostToDevice);
    for (int it = 0; it < 2; ++it) { int rc = dmet_knn_f32(x, ptr, B, N, D, k, nbr, dist, ws, wsb, 0); if (rc) { printf("rc=%d %s\n", rc, dmet_last_error()); return 1; } hipDeviceSynchronize(); }
    int blocks = (int)((N + 127) / 128) + B + 1024; if (blocks > 65536) blocks = 65536;
    { int live = 0; std::vector<unsigned long long> tmp((1 << 16) * 4); }
    std::vector<unsigned long long> st((1 << 16) * 4);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(dmet::g_knn_stamps), st.size() * 8);
    // drop workgroups that exited at once (no end stamp)
    { int w = 0; for (int b = 0; b < blocks; ++b) if (st[b*4+1] != 0) { for (int q = 0; q < 4; ++q) st[w*4+q] = st[b*4+q]; ++w; } blocks = w; }
    unsigned long long t0 = ~0ull, t1 = 0; for (int b = 0; b < blocks; ++b) { t0 = std::min(t0, st[b*4]); t1 = std::max(t1, st[b*4+1]); }
    printf("B=%d blocks=%d kernel span %.3f ms (100MHz realtime)\n", B, blocks, (t1 - t0) / 1e5);
    double sumlife = 0; std::map<unsigned, int> per_cu, per_simd, per_xcc;
    for (int b = 0; b < blocks; ++b) { sumlife += (st[b*4+1] - st[b*4]) / 1e5; unsigned hw = (unsigned)st[b*4+2], xcc = (unsigned)st[b*4+3] & 0xf;
        unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7; unsigned cuid = (xcc << 12) | (se << 8) | (sh << 4) | cu; per_cu[cuid]++; per_simd[(cuid << 2) | simd]++; per_xcc[xcc]++; }
    { std::vector<double> lf; for (int b = 0; b < blocks; ++b) lf.push_back((st[b*4+1]-st[b*4])/1e5); std::sort(lf.begin(), lf.end()); printf("lifetime ms: min %.3f p25 %.3f p50 %.3f p75 %.3f max %.3f\n", lf[0], lf[blocks/4], lf[blocks/2], lf[blocks*3/4], lf.back()); }
    { std::map<unsigned,double> busy; for (int b = 0; b < blocks; ++b) { unsigned hw = (unsigned)st[b*4+2], xcc = (unsigned)st[b*4+3] & 0xf; unsigned key = (xcc << 16) | (hw & 0xfff0); busy[key] = std::max(busy[key], (double)(st[b*4+1]-t0)/1e5); } std::vector<double> e; for (auto& kv : busy) e.push_back(kv.second); std::sort(e.begin(), e.end()); printf("per-SIMD finish time ms: min %.3f p10 %.3f p50 %.3f p90 %.3f max %.3f (n=%zu)\n", e[0], e[e.size()/10], e[e.size()/2], e[e.size()*9/10], e.back(), e.size()); }
    printf("mean wave lifetime %.3f ms; distinct CUs used %zu, distinct SIMDs %zu\n", sumlife / blocks, per_cu.size(), per_simd.size());
    std::map<int,int> h; for (auto& kv : per_simd) h[kv.second]++; printf("waves-per-SIMD histogram (over whole kernel): "); for (auto& kv : h) printf("%d:%d ", kv.first, kv.second); printf("\n");
    printf("per XCC: "); for (auto& kv : per_xcc) printf("%u:%d ", kv.first, kv.second); printf("\n");
    { std::vector<std::pair<double,int>> lf; for (int b = 0; b < blocks; ++b) lf.push_back({(st[b*4+1]-st[b*4])/1e5, b}); std::sort(lf.rbegin(), lf.rend()); printf("slowest waves (lifetime ms, compacted idx, start ms, hwid): "); for (int q = 0; q < 12; ++q) { int b = lf[q].second; printf("[%.2f #%d s%.2f hw%llx x%llu] ", lf[q].first, b, (st[b*4]-t0)/1e5, st[b*4+2] & 0xffff, st[b*4+3] & 0xf); } printf("\n"); }
    { double sum[8] = {0}, mx[8] = {0}; int cnt[8] = {0}; for (int b = 0; b < blocks; ++b) { int x = st[b*4+3] & 7; double l = (st[b*4+1]-st[b*4])/1e5; sum[x] += l; cnt[x]++; mx[x] = std::max(mx[x], l); } printf("per-XCC mean/max lifetime: "); for (int x = 0; x < 8; ++x) printf("%d: %.2f/%.2f  ", x, sum[x]/std::max(cnt[x],1), mx[x]); printf("\n"); }
    // concurrency profile: how many waves alive at 10 sample points
    for (int sidx = 0; sidx <= 10; ++sidx) { unsigned long long t = t0 + (t1 - t0) * sidx / 10; int alive = 0; for (int b = 0; b < blocks; ++b) if (st[b*4] <= t && st[b*4+1] >= t) alive++; printf("%d ", alive); } printf(" <- waves alive at 0..100%% of span\n");
    // start-time distribution
    std::vector<double> starts; for (int b = 0; b < blocks; ++b) starts.push_back((st[b*4] - t0) / 1e5); std::sort(starts.begin(), starts.end());
    printf("start times ms: p10 %.3f p50 %.3f p90 %.3f max %.3f\n", starts[blocks/10], starts[blocks/2], starts[blocks*9/10], starts.back());
    return 0;
}
