// Link stub for host_tests (CPU-only): the scorers reference hip_gamma_model's alpha accessors, but
// the unit tests never construct a GPU model.  Any attempt to do so aborts: there is no CPU path.
#include <cstdlib>

#include "cafe_host.h"

namespace cafe {
void hip_gamma_model::set_alpha(double) { std::abort(); }
}  // namespace cafe
