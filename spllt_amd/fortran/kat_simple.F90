!> The 3x3 known-answer case of the reference's example/C/simple.c, through the
!! Fortran API: tridiag(-1,2,-1) x = 1  ->  x = (1.5, 2, 1.5).  Exit code 0 on success.
program kat_simple
  use iso_c_binding
  use spllt_hip_mod
  implicit none
  type(spllt_akeep) :: akeep
  type(spllt_fkeep) :: fkeep
  type(spllt_options) :: options
  type(spllt_inform) :: info
  integer(c_int) :: ptr(4), row(5), order(3)
  real(c_double) :: val(5), x(3)
  ptr = (/ 1, 3, 5, 6 /)
  row = (/ 1, 2, 2, 3, 3 /)
  val = (/ 2.0d0, -1.0d0, 2.0d0, -1.0d0, 2.0d0 /)
  x = 1.0d0
  options%nb = 4
  call spllt_analyse(akeep, fkeep, options, 3, ptr, row, info, order)
  if (info%flag < 0) stop 1
  call spllt_factor(akeep, fkeep, options, val, info)
  if (info%flag < 0) stop 2
  call spllt_wait()
  call spllt_solve(fkeep, options, order, 1, x, info, 0)
  if (info%flag < 0) stop 3
  print '(a,3f12.8)', 'x =', x
  if (maxval(abs(x - (/ 1.5d0, 2.0d0, 1.5d0 /))) > 1d-14) stop 4
  call spllt_finalize(akeep, fkeep)
end program kat_simple
