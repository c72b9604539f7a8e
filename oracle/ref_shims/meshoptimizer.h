// oracle/ref_shims/meshoptimizer.h — TEST INFRASTRUCTURE.  meshoptimizer is an un-vendored submodule of the reference
// (.gitmodules:7-9).  src/scenecache.cpp calls its stream codecs only on the `compressed` branches; the pin built from
// this header (oracle/ref_scenecache.cpp) exercises the raw branches, so the codecs are declarations that abort when
// reached.  Signatures as used at src/scenecache.cpp:64-116,214-243.
#pragma once
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

#define NV_NO_MESHOPT(name) (fprintf(stderr, "oracle/_ref: %s needs meshoptimizer, which the reference does not vendor\n", name), abort(), 0)

inline size_t meshopt_encodeVertexBufferBound(size_t, size_t) { return NV_NO_MESHOPT("meshopt_encodeVertexBufferBound"); }
inline size_t meshopt_encodeVertexBufferLevel(unsigned char*, size_t, const void*, size_t, size_t, int) { return NV_NO_MESHOPT("meshopt_encodeVertexBufferLevel"); }
inline size_t meshopt_encodeIndexBufferBound(size_t, size_t) { return NV_NO_MESHOPT("meshopt_encodeIndexBufferBound"); }
inline size_t meshopt_encodeIndexBuffer(unsigned char*, size_t, const unsigned int*, size_t) { return NV_NO_MESHOPT("meshopt_encodeIndexBuffer"); }
inline size_t meshopt_encodeMeshletBound(size_t, size_t) { return NV_NO_MESHOPT("meshopt_encodeMeshletBound"); }
inline size_t meshopt_encodeMeshlet(unsigned char*, size_t, const unsigned int*, size_t, const unsigned char*, size_t) { return NV_NO_MESHOPT("meshopt_encodeMeshlet"); }
inline int meshopt_decodeVertexBuffer(void*, size_t, size_t, const unsigned char*, size_t) { return NV_NO_MESHOPT("meshopt_decodeVertexBuffer"); }
inline int meshopt_decodeIndexBuffer(unsigned int*, size_t, const unsigned char*, size_t) { return NV_NO_MESHOPT("meshopt_decodeIndexBuffer"); }
inline int meshopt_decodeMeshlet(void*, size_t, size_t, void*, size_t, size_t, const unsigned char*, size_t) { return NV_NO_MESHOPT("meshopt_decodeMeshlet"); }
