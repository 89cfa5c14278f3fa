// sg_mjcf.h -- native MJCF subset compiler (sg_mjcf.cpp): XML file -> model blob
#pragma once
#include <string>

// Compiles the scene at xml_path (includes resolved relative to its directory) into the tagged-array container of
// include/softgrip_model.h.  Returns false and sets *err for files outside the supported MJCF subset.
bool sg_mjcf_compile_file(const char* xml_path, bool composite_neighbors, bool implicit_tendon_damping, std::string* blob, std::string* err);
