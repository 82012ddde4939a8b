!> TEST INFRASTRUCTURE ONLY -- not part of the product path.
!!
!! bind(C) shim around the UNMODIFIED reference module `splpak_module`
!! (compiled from /root/reference/src/splpak.F90 where it lies; see
!! oracle/Makefile target `ref`).  It lets the Python test-suite and the
!! golden-vector generator (oracle/gen_golden.py) call the reference's public
!! generics `initialize` (=> splcc/splcw, src/splpak.F90:117) and `evaluate`
!! (=> splfe/splde, src/splpak.F90:118) through ctypes.
!!
!! Nothing here restates reference code: every routine only forwards its
!! arguments to the reference's type-bound procedures.
module splpak_ref_shim
    use iso_c_binding
    use splpak_module, only: splpak_type, splpak_wp
    implicit none
    private
    public :: ref_splcw, ref_splcc, ref_splde_many, ref_splfe_many, ref_wp_bytes

contains

    !> size in bytes of the reference's working precision (REAL32/REAL64 build)
    integer(c_int) function ref_wp_bytes() bind(C, name='ref_wp_bytes')
        ref_wp_bytes = int(storage_size(1.0_splpak_wp)/8, c_int)
    end function ref_wp_bytes

    !> reference `initialize` with weights (splcw, src/splpak.F90:512)
    subroutine ref_splcw(ndim, xdata, l1xdat, ydata, wdata, nwdata, ndata, xmin, xmax, &
                         nodes, xtrap, coef, ncf, work, nwrk, ierror) bind(C, name='ref_splcw')
        integer(c_int), value :: ndim, l1xdat, nwdata, ndata, ncf, nwrk
        real(splpak_wp), intent(in) :: xdata(l1xdat, *), ydata(*), wdata(nwdata)
        real(splpak_wp), intent(in) :: xmin(*), xmax(*)
        integer(c_int), intent(in) :: nodes(*)
        real(splpak_wp), value :: xtrap
        real(splpak_wp), intent(inout) :: coef(ncf), work(nwrk)
        integer(c_int), intent(out) :: ierror
        type(splpak_type) :: s
        integer :: ie, nd
        nd = max(ndim, 1)
        call s%initialize(int(ndim), xdata(1:l1xdat, 1:max(ndata,1)), int(l1xdat), &
                          ydata(1:max(ndata,1)), wdata, int(ndata), xmin(1:nd), xmax(1:nd), &
                          nodes(1:nd), xtrap, coef, int(ncf), work, int(nwrk), ie)
        ierror = int(ie, c_int)
        call s%destroy()
    end subroutine ref_splcw

    !> reference `initialize` without weights (splcc, src/splpak.F90:421)
    subroutine ref_splcc(ndim, xdata, l1xdat, ydata, ndata, xmin, xmax, &
                         nodes, xtrap, coef, ncf, work, nwrk, ierror) bind(C, name='ref_splcc')
        integer(c_int), value :: ndim, l1xdat, ndata, ncf, nwrk
        real(splpak_wp), intent(in) :: xdata(l1xdat, *), ydata(*)
        real(splpak_wp), intent(in) :: xmin(*), xmax(*)
        integer(c_int), intent(in) :: nodes(*)
        real(splpak_wp), value :: xtrap
        real(splpak_wp), intent(inout) :: coef(ncf), work(nwrk)
        integer(c_int), intent(out) :: ierror
        type(splpak_type) :: s
        integer :: ie, nd
        nd = max(ndim, 1)
        call s%initialize(int(ndim), xdata(1:l1xdat, 1:max(ndata,1)), int(l1xdat), &
                          ydata(1:max(ndata,1)), int(ndata), xmin(1:nd), xmax(1:nd), &
                          nodes(1:nd), xtrap, coef, int(ncf), work, int(nwrk), ie)
        ierror = int(ie, c_int)
        call s%destroy()
    end subroutine ref_splcc

    !> loop of scalar reference `evaluate` calls with nderiv (splde, src/splpak.F90:1089)
    subroutine ref_splde_many(ndim, nq, xq, ldxq, nderiv, coef, ncf, xmin, xmax, nodes, &
                              out, ierror) bind(C, name='ref_splde_many')
        integer(c_int), value :: ndim, nq, ldxq, ncf
        real(splpak_wp), intent(in) :: xq(ldxq, *), xmin(*), xmax(*)
        integer(c_int), intent(in) :: nderiv(*), nodes(*)
        real(splpak_wp), intent(inout) :: coef(ncf)
        real(splpak_wp), intent(out) :: out(*)
        integer(c_int), intent(out) :: ierror
        type(splpak_type) :: s
        integer :: i, ie, nd
        real(splpak_wp), allocatable :: x(:)
        nd = max(ndim, 1)
        allocate(x(nd))
        call s%destroy(nd)
        ierror = 0
        do i = 1, nq
            x(1:nd) = xq(1:nd, i)
            out(i) = s%evaluate(int(ndim), x, nderiv(1:nd), coef, xmin(1:nd), xmax(1:nd), &
                                nodes(1:nd), ie)
            if (ie /= 0) ierror = int(ie, c_int)
        end do
        call s%destroy()
    end subroutine ref_splde_many

    !> loop of scalar reference `evaluate` calls without nderiv (splfe, src/splpak.F90:1258)
    subroutine ref_splfe_many(ndim, nq, xq, ldxq, coef, ncf, xmin, xmax, nodes, &
                              out, ierror) bind(C, name='ref_splfe_many')
        integer(c_int), value :: ndim, nq, ldxq, ncf
        real(splpak_wp), intent(in) :: xq(ldxq, *), xmin(*), xmax(*)
        integer(c_int), intent(in) :: nodes(*)
        real(splpak_wp), intent(inout) :: coef(ncf)
        real(splpak_wp), intent(out) :: out(*)
        integer(c_int), intent(out) :: ierror
        type(splpak_type) :: s
        integer :: i, ie, nd
        real(splpak_wp), allocatable :: x(:)
        nd = max(ndim, 1)
        allocate(x(nd))
        call s%destroy(nd)
        ierror = 0
        do i = 1, nq
            x(1:nd) = xq(1:nd, i)
            out(i) = s%evaluate(int(ndim), x, coef, xmin(1:nd), xmax(1:nd), nodes(1:nd), ie)
            if (ie /= 0) ierror = int(ie, c_int)
        end do
        call s%destroy()
    end subroutine ref_splfe_many

end module splpak_ref_shim
