#pragma once
// SSAL_MEASURE: measurement-only code (phase ablation, environment-driven launcher knobs).  Never defined in the
// product build of libssal_hip.so (build.py); tools/phase_trace.py builds its private variant with it.
#if defined(SSAL_PHASE_TRACE) && !defined(SSAL_MEASURE)
#define SSAL_MEASURE 1
#endif
