/* amwg_rtc_headers.c -- the kernel headers as NUL-terminated text inside libamwg.so, so that
 * amwg_create_user can hand them to hiprtc (csrc/amwg_core.hip, compile_user).  Paths are
 * relative to this directory (the Makefile builds from here). */
#define AMWG_TEXT(sym, file)                                                        \
  __asm__(".section .rodata\n.global " #sym "\n.type " #sym ", @object\n" #sym ":\n" \
          ".incbin \"" file "\"\n.byte 0\n.size " #sym ", .-" #sym "\n.text\n")
AMWG_TEXT(amwg_hdr_stdint, "amwg_stdint.h");
AMWG_TEXT(amwg_hdr_types, "amwg_types.h");
AMWG_TEXT(amwg_hdr_math, "amwg_math.h");
AMWG_TEXT(amwg_hdr_div, "amwg_div.h");
AMWG_TEXT(amwg_hdr_ld, "amwg_ld.h");
AMWG_TEXT(amwg_hdr_philox, "amwg_philox.h");
AMWG_TEXT(amwg_hdr_kernel, "amwg_kernel.h");
AMWG_TEXT(amwg_hdr_user, "amwg_user.h");
AMWG_TEXT(amwg_hdr_twoval, "amwg_twoval.h");
AMWG_TEXT(amwg_hdr_kval, "amwg_kval.h");
AMWG_TEXT(amwg_hdr_trig, "amwg_trig.h");
AMWG_TEXT(amwg_hdr_pass, "amwg_pass.h");
AMWG_TEXT(amwg_hdr_rows, "amwg_rows.h");
AMWG_TEXT(amwg_hdr_window, "amwg_window.h");
AMWG_TEXT(amwg_hdr_ptail, "amwg_ptail.h");
