/*
 * The reference operator opens with two OpenACC runtime calls that pick a device
 * (openacc.cpp:64-68).  libgomp does define both, but its acc_init aborts the process
 * when asked for an NVIDIA device on a host that has none, and no compiler here
 * offloads OpenACC to AMD GPUs.  The reference build under oracle/_ref/ therefore
 * runs the loop nest as the serial host code it is when the pragmas are ignored, and
 * resolves the two device-selection calls to these no-ops.  Neither touches any data
 * or arithmetic.  This file is linked only into oracle/_ref/libref_openacc.so.
 */
void acc_init(int device_type) { (void)device_type; }
void acc_set_device_num(int device_num, int device_type) {
  (void)device_num;
  (void)device_type;
}
