#include "sf_common.h"
thread_local char sf_err_buf[512] = "";
extern "C" {
int sf_version(void) { return 100; }
const char* sf_last_error_string(void) { return sf_err_buf; }
}
