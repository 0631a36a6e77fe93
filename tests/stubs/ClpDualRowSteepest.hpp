// STUB of src/ClpDualRowSteepest.hpp:26-230 (what clpGpuDual reads from the model's pivot-rule object)
#ifndef ClpDualRowSteepest_STUB
#define ClpDualRowSteepest_STUB
#include "ClpDualRowPivot.hpp"
class ClpDualRowSteepest : public ClpDualRowPivot {
public:
  ClpDualRowSteepest(int mode = 3);
  inline int mode() const
  {
    return mode_;
  }
protected:
  int mode_;
};
#endif
