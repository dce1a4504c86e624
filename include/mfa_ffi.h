/*
 * mfa_ffi.h -- the header name the reference's callers include
 * (Sources/MFAFFI/include/mfa_ffi.h; bindgen input of examples/rust-ffi/build.rs:9-41).
 * The MI355X build keeps the name and forwards to umfa_abi.h, which declares the same
 * 29 prototypes plus the 13 symbols the reference exports without declaring.
 */
#ifndef MFA_FFI_H
#define MFA_FFI_H
#include "umfa_abi.h"
#endif
