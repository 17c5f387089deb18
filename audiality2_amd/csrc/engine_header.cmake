# The one generated file of the engine's include tree, include/audiality2.h, made the way the
# engine's own build makes it: cmake's configure_file() on the engine's template, version numbers
# read from the engine's top-level CMakeLists.txt.  Needed to compile a2amd_walk.c against the
# engine's internal headers (INTEGRATION.md option C).
# Usage: cmake -DENGINE=<engine source tree> -DOUT=<dir> -P engine_header.cmake
file(READ "${ENGINE}/CMakeLists.txt" _top)
foreach(_k MAJOR MINOR PATCH BUILD)
  string(REGEX MATCH "set\\(VERSION_${_k} ([0-9]+)\\)" _m "${_top}")
  set(VERSION_${_k} "${CMAKE_MATCH_1}")
endforeach()
configure_file("${ENGINE}/include/audiality2.h.cmake" "${OUT}/audiality2.h" @ONLY)
