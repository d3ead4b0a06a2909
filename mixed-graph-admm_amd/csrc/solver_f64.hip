// float64 engine (streaming kernels)
#include "engine.h"

EngineBase* mg_make_engine_f64(mgadmm_solver* s) { return new Engine<double>(s); }
