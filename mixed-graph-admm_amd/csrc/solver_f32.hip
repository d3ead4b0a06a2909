// float32 engine (streaming kernels + LDS-resident path)
#include "engine.h"

EngineBase* mg_make_engine_f32(mgadmm_solver* s) { return new Engine<float>(s); }
