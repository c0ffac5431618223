// placeholder, replaced below
#include "cpd_plan.h"
extern "C" int prg_cpd_mstep_nonrigid(prg_cpd* h, double lmd) {
    (void)h; (void)lmd;
    prg::set_error("prg_cpd_mstep_nonrigid: not built yet");
    return PRG_ERR_STATE;
}
