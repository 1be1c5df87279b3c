// device_stub.cpp -- stands in for ff_device.hip, ff_dev_stage.hip and ff_dev_run.hip in the sanitizer build of the HOST code
// (make asan): every device entry point fails with FF_ERR_DEVICE.  Never linked into the
// product library.
#include "ff_host.hpp"

namespace ff {
int device_count() { return 0; }
void device_warmup(int) {}
size_t device_free_bytes(int) { return 0; }
int unifrac_dists_info(const ff_problem *, const ff_options *, double *, ff_plan_info *, char *err, size_t errlen)
{
    return fail(FF_ERR_DEVICE, err, errlen, "device stub");
}
int unifrac_leaves_info(const ff_tree *, int64_t, const int64_t *, const int64_t *, const double *, int,
                        const ff_options *, double *, ff_plan_info *, char *err, size_t errlen, bool)
{
    return fail(FF_ERR_DEVICE, err, errlen, "device stub");
}
ShardRunner::ShardRunner(const ff_tree *tree, int64_t n, const int64_t *lp, const int64_t *li, const double *lv, int un,
                         const ff_options &opt)
    : tree_(tree), n_(n), lp_(lp), li_(li), lv_(lv), unnorm_(un), opt_(opt)
{
}
ShardRunner::~ShardRunner() {}
int ShardRunner::create(int32_t, int32_t, int, char *err, size_t errlen) { return fail(FF_ERR_DEVICE, err, errlen, "device stub"); }
int ShardRunner::run(int32_t, int32_t, double *, ff_plan_info *, char *err, size_t errlen)
{
    return fail(FF_ERR_DEVICE, err, errlen, "device stub");
}
int ShardRunner::run_device(int32_t, int32_t, const double **, int64_t *, ff_plan_info *, char *err, size_t errlen)
{
    return fail(FF_ERR_DEVICE, err, errlen, "device stub");
}
int ShardRunner::prepare(int32_t, int32_t, char *err, size_t errlen) { return fail(FF_ERR_DEVICE, err, errlen, "device stub"); }
int ShardRunner::device() const { return 0; }
struct TextPipeline::Impl {};
TextPipeline::TextPipeline(DistWriter *) : impl_(nullptr) {}
TextPipeline::~TextPipeline() {}
int TextPipeline::prepare(int64_t, char *err, size_t errlen) { return fail(FF_ERR_DEVICE, err, errlen, "device stub"); }
int TextPipeline::submit(int, const double *, int64_t, char *err, size_t errlen) { return fail(FF_ERR_DEVICE, err, errlen, "device stub"); }
void TextPipeline::abandon() {}
int TextPipeline::drain(char *err, size_t errlen) { return fail(FF_ERR_DEVICE, err, errlen, "device stub"); }
}  // namespace ff

extern "C" size_t ff_text_bound(int64_t n) { return n > 0 ? (size_t)n * 25 : 0; }
extern "C" int ff_format_distances_device(const double *, int64_t, char *, size_t *, void *, char *err, size_t errlen)
{
    return ff::fail(FF_ERR_DEVICE, err, errlen, "device stub");
}
extern "C" int ff_unifrac_dists(const ff_problem *, const ff_options *, double *, char *err, size_t errlen)
{
    return ff::fail(FF_ERR_DEVICE, err, errlen, "device stub");
}
