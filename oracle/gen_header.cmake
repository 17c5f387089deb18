# Runs cmake's own configure_file() on the reference's public header template
# (include/audiality2.h.cmake) exactly as the reference build would, without
# running the reference's build system.  Version numbers are read from the
# reference's top-level CMakeLists.txt so nothing is invented here.
# Usage: cmake -DREF=/root/reference -DOUT=_ref/include -P gen_header.cmake
file(READ "${REF}/CMakeLists.txt" _top)
foreach(_k MAJOR MINOR PATCH BUILD)
  string(REGEX MATCH "set\\(VERSION_${_k} ([0-9]+)\\)" _m "${_top}")
  set(VERSION_${_k} "${CMAKE_MATCH_1}")
endforeach()
configure_file("${REF}/include/audiality2.h.cmake" "${OUT}/audiality2.h" @ONLY)
