// TEST INFRASTRUCTURE ONLY: util/debug.h of the reference mentions cuRAND status codes; nothing here calls cuRAND.
#pragma once
typedef int curandStatus_t;
enum { CURAND_STATUS_SUCCESS = 0 };
