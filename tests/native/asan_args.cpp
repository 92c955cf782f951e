// AddressSanitizer run of the HOST side of libpacingpseudo_hip: pp_runtime.cpp and the argument checks
// of the launch wrappers, built for the host only (`make asan`, no GPU needed -- GPU ASAN is not available on this pool).
// Every call below must be rejected by its PP_CHECK_ARG / workspace checks BEFORE anything is launched: return code < 0,
// a message in pp_last_error(), no out-of-bounds access while building it.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "pacingpseudo_hip.h"

void pp_set_error(const char* fmt, ...);

static int failures = 0;
#define EXPECT_REJECT(call)                                                              \
  do {                                                                                   \
    const int rc = (call);                                                               \
    const char* msg = pp_last_error();                                                   \
    if (rc >= 0 || !msg || !msg[0]) { printf("NOT REJECTED: %s -> %d\n", #call, rc); ++failures; } \
  } while (0)

int main() {
  // thread-local error buffer: a message far longer than the buffer must be truncated, not overflow
  std::string big(5000, 'x');
  pp_set_error("%s %d", big.c_str(), 7);
  if (strlen(pp_last_error()) >= 512) { printf("error buffer not bounded\n"); ++failures; }
  if (pp_version() < 100) ++failures;
  // profiler bookkeeping without a device: collect with nothing recorded, select / enable toggles
  std::vector<double> out(15 * 5, -1.0);
  pp_prof_enable(0);
  pp_prof_select(~0ull);
  if (pp_prof_collect(out.data(), 15) != 0 || out[0] != 0.0) ++failures;
  // named ranges resolve lazily and must be harmless without a profiler attached
  pp_range_push("asan");
  pp_range_pop();
  pp_range_push(nullptr);

  float buf[64] = {0};
  double dbuf[64] = {0};
  int64_t lbuf[8] = {0};
  int ibuf[8] = {0};
  void* ws = buf;
  // null pointers / misaligned / inconsistent shapes: one representative per source file
  EXPECT_REJECT(pp_conv3x3_fwd(nullptr, 4, 4, buf, buf, buf, 4, 4, 1, 8, 8, 1, 0, nullptr));
  EXPECT_REJECT(pp_conv3x3_fwd(buf, 3, 3, buf, buf, buf, 4, 4, 1, 8, 8, 1, 0, nullptr));                 // C % 4
  EXPECT_REJECT(pp_conv3x3_fwd(buf + 1, 4, 4, buf, buf, buf, 4, 4, 1, 8, 8, 1, 0, nullptr));             // alignment
  EXPECT_REJECT(pp_conv3x3_wino_fwd(buf, 8, 8, buf, buf, buf, 8, 8, 1, 7, 8, 1, 0, nullptr, ws, 1 << 20, nullptr));   // H % 2 dil
  EXPECT_REJECT(pp_conv3x3_wino_fwd(buf, 8, 8, buf, buf, buf, 8, 8, 1, 8, 8, 1, 0, nullptr, ws, 16, nullptr));        // workspace
  EXPECT_REJECT(pp_conv3x3_wino_fwd_f16x3(buf, 12, 12, buf, buf, buf, 8, 8, 1, 8, 8, 1, 0, nullptr, ws, 1 << 24, nullptr));  // K % 8
  EXPECT_REJECT(pp_wino_pack_weights_f16x3(buf, 8, 12, 4, buf, nullptr, nullptr));                       // K % 8
  EXPECT_REJECT(pp_bn_lrelu_bwd_eval(nullptr, 4, buf, 4, buf, buf, buf, buf, 4, buf, buf, buf, 0, 4, 16, 0.01f, ws, 1 << 20, nullptr, nullptr));
  EXPECT_REJECT(pp_bn_lrelu_bwd_eval(buf, 4, buf, 4, buf, buf, buf, buf, 4, buf, buf, buf, 0, 4, 16, 0.01f, ws, 8, nullptr, nullptr));   // workspace
  EXPECT_REJECT(pp_maxpool2_fwd(buf, 4, buf, 4, 4, 1, 7, 8, nullptr));                                    // odd H
  EXPECT_REJECT(pp_argmax_channels(nullptr, 1, 6, 64, lbuf, nullptr));
  EXPECT_REJECT(pp_adam_step(buf, buf, nullptr, buf, 16, 1e-4f, 0.9f, 0.999f, 1e-8f, 3e-4f, 1, nullptr));
  EXPECT_REJECT(pp_adam_step(buf, buf, buf, buf, 16, 1e-4f, 0.9f, 0.999f, 1e-8f, 3e-4f, 0, nullptr));   // step counter starts at 1
  EXPECT_REJECT(pp_aug_scalar_map(nullptr, 1, 8, 8, buf, ibuf, nullptr));
  EXPECT_REJECT(pp_aug_add_field(buf, nullptr, 1, 8, 8, ibuf, nullptr));
  EXPECT_REJECT(pp_aug_coef(dbuf, nullptr, buf, 3, 1, buf, nullptr));                                    // mode 3 needs stats0
  if (failures) { printf("%d failure(s)\n", failures); return 1; }
  printf("asan_args: all argument checks rejected their input, no sanitizer report\n");
  return 0;
}
