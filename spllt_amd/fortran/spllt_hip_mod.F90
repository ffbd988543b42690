!> Fortran API of the spllt-hip engine.
!!
!! Keeps the call shapes of the reference's user-level routines
!! (spllt_analyse  src/spllt_analyse_mod.F90:23,
!!  spllt_factor   src/spllt_mod.F90:141,  spllt_wait  src/spllt_mod.F90:172,
!!  spllt_solve    src/spllt_solve_mod.F90:8-12)
!! and forwards to the C-ABI of libspllt_hip.so (include/spllt_iface.h).
!! The opaque akeep/fkeep pair replaces the reference's derived types
!! spllt_akeep / spllt_fkeep (src/spllt_data_mod.F90:315-388): their content
!! lives in the engine (symbolic structure on the host, L in HBM).
module spllt_hip_mod
  use iso_c_binding
  implicit none
  private
  public :: spllt_options, spllt_inform, spllt_akeep, spllt_fkeep
  public :: spllt_analyse, spllt_factor, spllt_wait, spllt_solve, spllt_finalize

  integer, parameter :: wp = c_double

  !> mirror of spllt_options_t (include/spllt_iface.h:14-31); defaults of
  !! type spllt_options (src/spllt_data_mod.F90:260-286)
  type, bind(C) :: spllt_options
     integer(c_int) :: print_level = 0
     integer(c_int) :: nrhs = 1
     integer(c_int) :: ncpu = 1
     integer(c_int) :: nb = 256
     integer(c_int) :: nemin = 32
     integer(c_int) :: prune_tree = 1
     integer(c_int) :: min_width_blas = 8
     integer(c_int) :: nb_min = 32
     integer(c_int) :: nb_max = 32
     integer(c_int) :: nrhs_min = 1
     integer(c_int) :: nrhs_max = 1
     integer(c_int) :: nb_linear_comp = 0
     integer(c_int) :: nrhs_linear_comp = 0
     integer(c_int) :: chunk = 10
  end type spllt_options

  !> mirror of spllt_inform_t (include/spllt_iface.h:49-57)
  type, bind(C) :: spllt_inform
     integer(c_int) :: flag = 0
     integer(c_int) :: maxdepth = 0
     integer(c_int) :: num_factor = 0
     integer(c_int) :: num_flops = 0
     integer(c_int) :: num_nodes = 0
     integer(c_int) :: stat = 0
  end type spllt_inform

  type :: spllt_akeep
     type(c_ptr) :: h = c_null_ptr
  end type spllt_akeep
  type :: spllt_fkeep
     type(c_ptr) :: h = c_null_ptr
  end type spllt_fkeep

  interface
     subroutine c_analyse(akeep, fkeep, options, n, ptr, row, info, order) bind(C, name="spllt_analyse")
       import :: c_ptr, c_int, spllt_options, spllt_inform
       type(c_ptr) :: akeep, fkeep
       type(spllt_options) :: options
       integer(c_int), value :: n
       integer(c_int) :: ptr(*), row(*), order(*)
       type(spllt_inform) :: info
     end subroutine c_analyse
     subroutine c_factor(akeep, fkeep, options, nnz, val, info) bind(C, name="spllt_factor")
       import :: c_ptr, c_int, c_double, spllt_options, spllt_inform
       type(c_ptr), value :: akeep, fkeep
       type(spllt_options) :: options
       integer(c_int), value :: nnz
       real(c_double) :: val(*)
       type(spllt_inform) :: info
     end subroutine c_factor
     subroutine c_wait() bind(C, name="spllt_wait")
     end subroutine c_wait
     subroutine c_solve(fkeep, options, order, nrhs, x, info, job) bind(C, name="spllt_solve")
       import :: c_ptr, c_int, c_double, spllt_options, spllt_inform
       type(c_ptr), value :: fkeep
       type(spllt_options) :: options
       integer(c_int) :: order(*)
       integer(c_int), value :: nrhs, job
       real(c_double) :: x(*)
       type(spllt_inform) :: info
     end subroutine c_solve
     subroutine c_free_fkeep(fkeep, stat) bind(C, name="spllt_deallocate_fkeep")
       import :: c_ptr, c_int
       type(c_ptr) :: fkeep
       integer(c_int) :: stat
     end subroutine c_free_fkeep
     subroutine c_free_akeep(akeep, stat) bind(C, name="spllt_deallocate_akeep")
       import :: c_ptr, c_int
       type(c_ptr) :: akeep
       integer(c_int) :: stat
     end subroutine c_free_akeep
  end interface

contains

  !> spllt_analyse(akeep, fkeep, options, n, ptr, row, info, order):
  !! 1-based CSC of the lower triangle, as in the reference.
  subroutine spllt_analyse(akeep, fkeep, options, n, ptr, row, info, order)
    type(spllt_akeep), intent(inout) :: akeep
    type(spllt_fkeep), intent(inout) :: fkeep
    type(spllt_options), intent(inout) :: options
    integer, intent(in) :: n
    integer(c_int), intent(in) :: ptr(:), row(:)
    type(spllt_inform), intent(out) :: info
    integer(c_int), intent(out) :: order(:)
    call c_analyse(akeep%h, fkeep%h, options, int(n, c_int), ptr, row, info, order)
  end subroutine spllt_analyse

  !> spllt_factor(akeep, fkeep, options, val, info): asynchronous; call spllt_wait.
  subroutine spllt_factor(akeep, fkeep, options, val, info)
    type(spllt_akeep), intent(in) :: akeep
    type(spllt_fkeep), intent(inout) :: fkeep
    type(spllt_options), intent(inout) :: options
    real(wp), intent(in) :: val(:)
    type(spllt_inform), intent(out) :: info
    call c_factor(akeep%h, fkeep%h, options, int(size(val), c_int), val, info)
  end subroutine spllt_factor

  subroutine spllt_wait()
    call c_wait()
  end subroutine spllt_wait

  !> spllt_solve(fkeep, options, order, nrhs, x, info, job): job 0 both, 1 fwd, 2 bwd.
  subroutine spllt_solve(fkeep, options, order, nrhs, x, info, job)
    type(spllt_fkeep), intent(inout) :: fkeep
    type(spllt_options), intent(inout) :: options
    integer(c_int), intent(in) :: order(:)
    integer, intent(in) :: nrhs
    real(wp), intent(inout) :: x(*)
    type(spllt_inform), intent(out) :: info
    integer, intent(in), optional :: job
    integer(c_int) :: j
    j = 0
    if (present(job)) j = int(job, c_int)
    call c_solve(fkeep%h, options, order, int(nrhs, c_int), x, info, j)
  end subroutine spllt_solve

  subroutine spllt_finalize(akeep, fkeep)
    type(spllt_akeep), intent(inout) :: akeep
    type(spllt_fkeep), intent(inout) :: fkeep
    integer(c_int) :: st
    call c_free_fkeep(fkeep%h, st)
    call c_free_akeep(akeep%h, st)
  end subroutine spllt_finalize

end module spllt_hip_mod
