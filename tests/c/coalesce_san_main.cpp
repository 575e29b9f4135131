// coalesce_san_main.cpp — the scenarios of tests/test_coalesce_cpu.py as one executable, for the sanitizer builds
// (tests/c/Makefile: -fsanitize=thread, and -fsanitize=address,undefined).  1 / 8 / 64 / 256 callers, 1 / 2 / 4 lanes, with the
// leader holding its group open for returning callers and without (QV_COALESCE_LINGER_DIV is read once per process, so "linger off"
// is a think time far beyond the bound: nobody is known to be on the way), every 7th query taking the slow second pass (the early
// round — the hand-over the round-5 advisor found a race in).  Exit status: 0 clean, 1 a caller received a wrong share.
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
extern "C" int coalesce_harness(int lanes, unsigned max_group, unsigned n_threads, unsigned calls_per_thread, unsigned pass_us, unsigned think_us, int two_keys, unsigned slow_every,
                                unsigned long long* out);
int main(int argc, char** argv) {
    const unsigned scale = argc > 1 ? (unsigned)atoi(argv[1]) : 1;      // calls per thread multiplier
    unsigned long long bad = 0;
    const unsigned threads[] = {1, 8, 64, 256};
    const int lanes[] = {1, 2, 4};
    for (unsigned t : threads)
        for (int l : lanes)
            for (unsigned think : {0u, 3000u}) {
                if (think && t > 8) continue;                           // (think-time runs are about the linger bound, not about crowds)
                for (unsigned slow : {0u, 7u}) {
                    unsigned long long o[8];
                    const unsigned calls = (t >= 64 ? 6 : 20) * scale;
                    coalesce_harness(l, t >= 64 ? 256 : 64, t, calls, 300, think, 1, slow, o);
                    const bool ok = o[6] == 0 && o[0] + o[1] + o[2] == (unsigned long long)t * calls && o[7] <= (unsigned long long)l;
                    printf("callers %3u lanes %d think %4u us slow-every %u: solo %llu led %llu rode %llu groups %llu wrong %llu max passes %llu%s\n",
                           t, l, think, slow, o[0], o[1], o[2], o[3], o[6], o[7], ok ? "" : "  <-- FAILED");
                    if (!ok) bad++;
                }
            }
    return bad ? 1 : 0;
}
