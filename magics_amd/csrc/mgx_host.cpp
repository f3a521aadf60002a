// mgx_host.cpp — host-only helpers of the C ABI (no device needed): the GBP iteration schedules
// (crates/gbp_schedule/src/schedules/*.rs) and the variable-timestep rule
// (crates/magics/src/utils.rs:35-75) the driver uses to build a robot's graph.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/mgx.h"

namespace {

void reverse(uint8_t *s, int n) {
    for (int i = 0, j = n - 1; i < j; i++, j--) {
        uint8_t t = s[i];
        s[i] = s[j];
        s[j] = t;
    }
}
void fill_every(uint8_t *s, int len, int period) {
    for (int i = 0; i < len; i++) s[i] = (i % period) == 0;
}

// InterleaveEvenlyIter::recurse (interleave_evenly.rs:41-104): spread n trues over slice s
void interleave(uint8_t *s, int len, int n) {
    const int half = len / 2;
    if (n == len) { memset(s, 1, (size_t)len); return; }
    if (n == 0) { memset(s, 0, (size_t)len); return; }
    const bool n_odd = n & 1, len_odd = len & 1;
    if (n_odd && len_odd) {
        if (len % n == 0) { fill_every(s, len, len / n); return; }
        interleave(s, half, n / 2);
        s[half] = 1;
        interleave(s + half + 1, len - half - 1, n / 2);
        reverse(s + half + 1, len - half - 1);
    } else if (!n_odd && len_odd) {
        interleave(s, half, n / 2);
        reverse(s, half);
        s[half] = 0;
        interleave(s + half + 1, len - half - 1, n / 2);
    } else if (!n_odd && !len_odd) {
        if (len % n == 0) { fill_every(s, len, len / n); return; }
        interleave(s, half, n / 2);
        interleave(s + half, len - half, n / 2);
    } else {
        interleave(s, half, n / 2 + 1);
        reverse(s, half);
        interleave(s + half, len - half, n / 2);
    }
}

// one boolean stream of length `max` with exactly n trues
void stream(int kind, int n, int max, uint8_t *s) {
    switch (kind) {
    case MGX_SCHEDULE_CENTERED: {  // centered.rs:19-48
        const int mid = max / 2, hn = n / 2;
        const int start = mid >= hn ? mid - hn : 0;
        const int end = (start + n <= max) ? start + n - 1 : max - 1;
        for (int i = 0; i < max; i++) s[i] = (i >= start && i <= end);
        break;
    }
    case MGX_SCHEDULE_SOON_AS_POSSIBLE:  // soon_as_possible.rs:27-51
        for (int i = 0; i < max; i++) s[i] = i < n;
        break;
    case MGX_SCHEDULE_LATE_AS_POSSIBLE:  // late_as_possible.rs:30-48
        for (int i = 0; i < max; i++) s[i] = i >= max - n;
        break;
    case MGX_SCHEDULE_INTERLEAVE_EVENLY: interleave(s, max, n); break;
    case MGX_SCHEDULE_HALF_BEGINNING_HALF_END: {  // half_beginning_half_end.rs:19-43
        const int hn = n / 2, rem = n % 2;
        for (int i = 0; i < max; i++) s[i] = (i < hn || i >= max - hn - rem);
        break;
    }
    }
}

}  // namespace

extern "C" {

int mgx_schedule(int32_t kind, uint8_t n_internal, uint8_t n_external, uint8_t *steps, uint32_t capacity) {
    const int max = n_internal > n_external ? n_internal : n_external;
    if (kind < 0 || kind > 4 || !steps || (int)capacity < max) return MGX_ERR_INVALID;
    std::vector<uint8_t> a((size_t)max + 1), b((size_t)max + 1);
    stream(kind, n_internal, max, a.data());
    stream(kind, n_external, max, b.data());
    for (int i = 0; i < max; i++) steps[i] = (uint8_t)((a[i] ? MGX_STEP_INTERNAL : 0) | (b[i] ? MGX_STEP_EXTERNAL : 0));
    return max;
}

int mgx_variable_timesteps(uint32_t lookahead_horizon, uint32_t lookahead_multiple, uint32_t *timesteps, uint32_t capacity) {
    if (!timesteps || lookahead_multiple == 0) return MGX_ERR_INVALID;
    const float h = (float)lookahead_horizon, m = (float)lookahead_multiple;
    const uint32_t n = 1u + (uint32_t)(0.5f * (-1.0f + sqrtf(1.0f + 8.0f * h / m)));
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < lookahead_multiple * (n + 1u); i++) {
        const float section = (float)(i / lookahead_multiple);
        const float f = fmaf(m / 2.0f, section, fmaf(section, -m, (float)i)) * (section + 1.0f);
        if (cnt >= capacity) return MGX_ERR_INVALID;
        if (f >= h) {
            timesteps[cnt++] = lookahead_horizon;
            break;
        }
        timesteps[cnt++] = (uint32_t)f;
    }
    return (int)cnt;
}

}  // extern "C"
