// TEMPORARY stubs for group 3 until the kernel-description compiler lands (replaced next commit).
#include "../eg_internal.hpp"
#define STUB(sig) extern "C" int sig { eg::set_error("group 3 (model) not built yet"); return EG_ERR_UNSUPPORTED; }
STUB(eg_model_compile(eg_ctx*, const char*, eg_model**))
STUB(eg_model_free(eg_model*))
extern "C" const char* eg_model_plan_text(eg_model*) { return ""; }
extern "C" int eg_model_kernel_count(eg_model*, const char*) { return -1; }
extern "C" int eg_model_tensor_count(eg_model*) { return 0; }
STUB(eg_model_param_info(eg_model*, int, int*, int*, int64_t*, char*, size_t))
STUB(eg_model_param_write(eg_model*, int, const float*, int64_t))
STUB(eg_model_param_read(eg_model*, int, float*, int64_t))
STUB(eg_model_grad_bucket(eg_model*, const char*, float**, int64_t*))
STUB(eg_model_param_ptr(eg_model*, int, float**, int64_t*))
STUB(eg_model_set_input_host(eg_model*, const char*, const float*, int, const int64_t*))
STUB(eg_model_set_input_device(eg_model*, const char*, const float*, int, const int64_t*))
STUB(eg_model_run(eg_model*, const char*))
STUB(eg_model_run_backward(eg_model*, const char*))
STUB(eg_model_run_update(eg_model*, const char*))
STUB(eg_model_set_grad_scale(eg_model*, float))
STUB(eg_model_output_shape(eg_model*, const char*, int*, int64_t*))
STUB(eg_model_read_output(eg_model*, const char*, float*, int64_t))
STUB(eg_model_tensor_shape(eg_model*, int, int*, int64_t*))
STUB(eg_model_read_tensor(eg_model*, int, float*, int64_t))
STUB(eg_model_tensor_ptr(eg_model*, int, float**, int64_t*))
STUB(eg_model_set_epoch(eg_model*, int64_t))
extern "C" int64_t eg_model_epoch(eg_model*) { return 0; }
