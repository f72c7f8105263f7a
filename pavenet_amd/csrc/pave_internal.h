/* Shared between the translation units of libpave_hip.so (not part of the C ABI). */
#ifndef PAVE_INTERNAL_H_
#define PAVE_INTERNAL_H_
int pave_internal_fail(int code, const char* msg); /* records pave_last_error(), returns code */
#endif
