// Host CPUs: how many this process may use, and how its threads wait for the device (spin or nap).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <sched.h>
#include <sys/prctl.h>
#include <time.h>

namespace np2h {

// CPUs this process may actually use: the cgroup's CFS quota where there is one (a container that sees 256 hardware
// threads may be allowed 16 CPUs' worth of time: threads beyond the quota are throttled, not run), the affinity mask,
// else the hardware's count.
inline unsigned usable_cpus() {
    static const unsigned n = [] {
        unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) hw = std::min<unsigned>(hw, (unsigned)std::max(1, CPU_COUNT(&set)));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota|max> <period>"
            char q[64];
            unsigned long long per = 0;
            if (fscanf(f, "%63s %llu", q, &per) == 2 && per && strcmp(q, "max") != 0)
                hw = std::min<unsigned>(hw, (unsigned)std::max<unsigned long long>(1, (strtoull(q, nullptr, 10) + per - 1) / per));
            fclose(f);
        } else {
            long long quota = -1, per = 0; // cgroup v1
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (fscanf(g, "%lld", &quota) != 1) quota = -1;
                fclose(g);
            }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(g, "%lld", &per) != 1) per = 0;
                fclose(g);
            }
            if (quota > 0 && per > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (quota + per - 1) / per));
        }
        return hw;
    }();
    return n;
}

// How a host thread waits for the device to post to a host-mapped mailbox.  Spinning answers within a microsecond and
// costs one CPU per waiting thread — four batch groups of one rank keep ~7 CPUs busy (bench.py's host_cpu) —, which is
// the right trade while the process has the CPUs: with fewer than 8 usable CPUs per local rank (8 ranks of a node in a
// container whose quota is 16 CPUs) the spinners would eat the quota the recording threads need and the whole cgroup
// gets throttled.  Then a waiter spins for a few microseconds and goes on in naps of ~25 us (timer slack lowered for
// the thread): seven waits per step and group get ~15 us longer each, the CPUs stay free.
//   NP2_WAIT=spin|nap overrides; LOCAL_WORLD_SIZE (torchrun) = ranks sharing this host's CPUs.
inline bool wait_naps() {
    static const bool naps = [] {
        if (const char *e = getenv("NP2_WAIT")) return e[0] == 'n' || (e[0] == 's' && e[1] == 'l');
        unsigned ranks = 1;
        if (const char *e = getenv("LOCAL_WORLD_SIZE")) ranks = (unsigned)std::max(1, atoi(e));
        return usable_cpus() < 8u * ranks;
    }();
    return naps;
}
struct HostWait { // one wait: call pause() between polls
    uint32_t polls = 0;
    inline void pause() {
        if (!wait_naps() || ++polls < 256) {
            __builtin_ia32_pause();
            return;
        }
        static thread_local bool slack_set = false;
        if (!slack_set) {
            (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
            slack_set = true;
        }
        const timespec ts{0, 20000};
        (void)nanosleep(&ts, nullptr);
    }
};

// ranks of this job that share the host's CPUs (torchrun's LOCAL_WORLD_SIZE; 1 outside a launcher)
inline unsigned local_ranks() {
    static const unsigned n = [] {
        const char *e = getenv("LOCAL_WORLD_SIZE");
        return (unsigned)std::max(1, e ? atoi(e) : 1);
    }();
    return n;
}

} // namespace np2h
