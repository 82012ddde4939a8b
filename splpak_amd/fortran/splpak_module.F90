!> Drop-in `splpak_module` for MI355X: the public surface of jacobwilliams/splpak
!! (`splpak_wp`, `splpak_type` with the generics `initialize` => splcc/splcw,
!! `evaluate` => splfe/splde and `destroy`; reference src/splpak.F90:43-127) over the
!! HIP library `libsplpak_hip.so` (C ABI in include/splpak_hip.h).
!!
!! Argument lists, `ierror` codes and the text printed on `output_unit` follow the
!! reference (splcw :512-513, splcc :421-422, splde :1089, splfe :1258, cfaerr :399-407).
!! All numerical work is done on the GPU; there is no host fallback: without a GPU
!! the calls fail with a negative `ierror` and the library's message is printed.
!!
!! Differences a caller can observe (documented in INTEGRATION.md):
!!  * `work` is not used as scratch.  The reference's 106 check on `nwrk` is kept; the
!!    larger requirement of the dense solver (error 107 via suprls 32, :1443-1454) is
!!    not enforced, because the GPU path does not need the ncol*(ncol+1) array (550 GB
!!    at 64^3 nodes).  After a successful fit with xtrap /= 0 `work(1:ncol)` holds the
!!    sparse-area histogram exactly as the reference leaves it (:879-907).
!!  * additive: `evaluate_many` evaluates a batch of points in one kernel launch,
!!    `evaluate_derivatives` value + gradient (+ Hessian) of a batch in one pass.
!!
!! Build with -DREAL32 for single precision storage (as the reference, :33-41);
!! REAL128 has no GPU path and is rejected at compile time.
module splpak_module

    use iso_c_binding
    use iso_fortran_env, only: real32, real64, output_unit

    implicit none

    private

#ifdef REAL32
    integer,parameter :: wp = real32
#elif REAL128
#error "REAL128 has no MI355X path; build the reference for quad precision"
#else
    integer,parameter :: wp = real64
#endif

    integer,parameter,public :: splpak_wp = wp   !! working precision

    type,public :: splpak_type
        private
        integer :: mdim = 0    !! dimension of the last call (the reference keeps scratch here, :95-111)
    contains
        private
        generic,public   :: initialize    => splcc, splcw        !! fit
        generic,public   :: evaluate      => splfe, splde        !! one point
        generic,public   :: evaluate_many => splfe_many, splde_many  !! batch of points (additive)
        procedure,public :: evaluate_derivatives => splpak_derivs_many !! value + gradient (+ Hessian) of a batch (additive)
        procedure,public :: destroy       => destroy_splpak
        procedure,private :: splcc
        procedure,private :: splcw
        procedure,private :: splfe
        procedure,private :: splde
        procedure,private :: splfe_many
        procedure,private :: splde_many
        procedure,private :: splpak_derivs_many
    end type splpak_type

    interface
#ifdef REAL32
        integer(c_int32_t) function c_fit(ndim,xdata,l1xdat,ydata,wdata,ndata,xmin,xmax,nodes,xtrap,&
                                          coef,ncf,nwrk,hist,info) bind(C,name='splpak_fit_f32')
#else
        integer(c_int32_t) function c_fit(ndim,xdata,l1xdat,ydata,wdata,ndata,xmin,xmax,nodes,xtrap,&
                                          coef,ncf,nwrk,hist,info) bind(C,name='splpak_fit_f64')
#endif
            import :: c_int32_t, c_int64_t, c_ptr, wp
            integer(c_int32_t),value :: ndim, l1xdat
            integer(c_int64_t),value :: ndata, ncf, nwrk
            type(c_ptr),value :: xdata, ydata, wdata, xmin, xmax, nodes, coef, hist, info
            real(wp),value :: xtrap
        end function c_fit
#ifdef REAL32
        integer(c_int32_t) function c_eval(ndim,nq,xq,ldxq,nderiv,coef,xmin,xmax,nodes,out) &
                                           bind(C,name='splpak_eval_f32')
#else
        integer(c_int32_t) function c_eval(ndim,nq,xq,ldxq,nderiv,coef,xmin,xmax,nodes,out) &
                                           bind(C,name='splpak_eval_f64')
#endif
            import :: c_int32_t, c_int64_t, c_ptr
            integer(c_int32_t),value :: ndim, ldxq
            integer(c_int64_t),value :: nq
            type(c_ptr),value :: xq, nderiv, coef, xmin, xmax, nodes, out
        end function c_eval
#ifdef REAL32
        integer(c_int32_t) function c_eval_derivs(ndim,nq,xq,ldxq,order,coef,xmin,xmax,nodes,out,ldout) &
                                                  bind(C,name='splpak_eval_derivs_f32')
#else
        integer(c_int32_t) function c_eval_derivs(ndim,nq,xq,ldxq,order,coef,xmin,xmax,nodes,out,ldout) &
                                                  bind(C,name='splpak_eval_derivs_f64')
#endif
            import :: c_int32_t, c_int64_t, c_ptr
            integer(c_int32_t),value :: ndim, ldxq, order, ldout
            integer(c_int64_t),value :: nq
            type(c_ptr),value :: xq, coef, xmin, xmax, nodes, out
        end function c_eval_derivs
        integer(c_int32_t) function c_last_error(buf,buflen) bind(C,name='splpak_last_error_message')
            import :: c_int32_t, c_char
            character(kind=c_char) :: buf(*)
            integer(c_int32_t),value :: buflen
        end function c_last_error
    end interface

    contains

    !> Release the object's state (the reference reallocates scratch here, :136-165).
    subroutine destroy_splpak(me,ndim)
        class(splpak_type),intent(inout) :: me
        integer,intent(in),optional :: ndim
        me%mdim = 0
        if (present(ndim)) me%mdim = ndim
    end subroutine destroy_splpak

    !> ` IERR=nnnnn` + message on output_unit, as the reference's cfaerr (:399-407).
    subroutine report(ierr,mess)
        integer,intent(in) :: ierr
        character(len=*),intent(in) :: mess
        if (ierr /= 0) write (output_unit,'(A,I5)') ' IERR=', ierr
        write (output_unit,'(A)') trim(mess)
    end subroutine report

    !> a negative status is an infrastructure failure (no GPU, out of device memory, ...)
    subroutine report_library_failure(ierr,who)
        integer,intent(in) :: ierr
        character(len=*),intent(in) :: who
        character(kind=c_char) :: buf(512)
        character(len=512) :: msg
        integer :: n, i
        n = c_last_error(buf, 512_c_int32_t)
        msg = ''
        do i = 1, min(n,511)
            msg(i:i) = buf(i)
        end do
        call report(ierr, ' '//who//' - HIP library failure: '//trim(msg))
    end subroutine report_library_failure

    subroutine report_fit(ierr)
        integer,intent(in) :: ierr
        select case (ierr)
        case (101); call report(ierr,' splcc or splcw - NDIM is less than 1')
        case (102); call report(ierr,' splcc or splcw - NODES(IDIM) is less than 4 for some IDIM')
        case (103); call report(ierr,' splcc or splcw - XMIN(IDIM) equals XMAX(IDIM) for some IDIM')
        case (104); call report(ierr,' splcc or splcw - NCF (size of COEF) is too small')
        case (105); call report(ierr,' splcc or splcw - Ndata Is less than 1')
        case (106); call report(ierr,' splcc or splcw - NWRK (size of WORK) is too small')
        case (107); call report(ierr,' splcc or splcw - suprls failure '//&
                                     '(this usually indicates insufficient input data)')
        end select
    end subroutine report_fit

    !> Unweighted fit; same arguments as the reference's splcc (:421-422).
    subroutine splcc(me,ndim,xdata,l1xdat,ydata,ndata,xmin,xmax,nodes, &
                     xtrap,coef,ncf,work,nwrk,ierror)
        class(splpak_type),intent(inout) :: me
        integer,intent(in) :: ndim, l1xdat, ncf, nwrk, ndata
        real(wp),intent(in),target :: xdata(l1xdat,*)
        real(wp),intent(in),target :: ydata(*)
        real(wp),intent(in),target :: xmin(*), xmax(*)
        real(wp),intent(in) :: xtrap
        integer,intent(in),target :: nodes(*)
        real(wp),target :: work(*)
        real(wp),intent(out),target :: coef(*)
        integer,intent(out) :: ierror
        call fit_common(me,ndim,c_loc(xdata),l1xdat,c_loc(ydata),c_null_ptr,ndata,c_loc(xmin), &
                        c_loc(xmax),nodes,xtrap,coef,ncf,work,nwrk,ierror)
    end subroutine splcc

    !> Weighted fit; same arguments as the reference's splcw (:512-513).
    !! `wdata(1) < 0` means "no weights" exactly as in the reference (:581-588).
    subroutine splcw(me,ndim,xdata,l1xdat,ydata,wdata,ndata,xmin,xmax, &
                     nodes,xtrap,coef,ncf,work,nwrk,ierror)
        class(splpak_type),intent(inout) :: me
        integer,intent(in) :: ndim, l1xdat, ncf, nwrk, ndata
        real(wp),intent(in),target :: xdata(l1xdat,*)
        real(wp),intent(in),target :: ydata(*)
        real(wp),intent(in) :: wdata(:)
        real(wp),intent(in),target :: xmin(*), xmax(*)
        real(wp),intent(in) :: xtrap
        integer,intent(in),target :: nodes(*)
        real(wp),target :: work(*)
        real(wp),intent(out),target :: coef(*)
        integer,intent(out) :: ierror
        real(wp),allocatable,target :: wcopy(:)
        logical :: weighted
        weighted = .false.
        if (size(wdata) >= 1) weighted = wdata(1) >= 0.0_wp
        if (weighted .and. ndata >= 1) then
            allocate(wcopy(ndata))                 ! contiguous copy of the assumed-shape dummy
            wcopy(1:ndata) = wdata(1:ndata)
            call fit_common(me,ndim,c_loc(xdata),l1xdat,c_loc(ydata),c_loc(wcopy),ndata,c_loc(xmin), &
                            c_loc(xmax),nodes,xtrap,coef,ncf,work,nwrk,ierror)
        else
            call fit_common(me,ndim,c_loc(xdata),l1xdat,c_loc(ydata),c_null_ptr,ndata,c_loc(xmin), &
                            c_loc(xmax),nodes,xtrap,coef,ncf,work,nwrk,ierror)
        end if
    end subroutine splcw

    subroutine fit_common(me,ndim,xdata,l1xdat,ydata,wdata,ndata,xmin,xmax,nodes,xtrap,coef,ncf, &
                          work,nwrk,ierror)
        class(splpak_type),intent(inout) :: me
        integer,intent(in) :: ndim, l1xdat, ncf, nwrk, ndata
        type(c_ptr),intent(in) :: xdata, ydata, wdata, xmin, xmax
        integer,intent(in),target :: nodes(*)
        real(wp),intent(in) :: xtrap
        real(wp),target :: coef(*), work(*)
        integer,intent(out) :: ierror
        integer(c_int32_t) :: rc
        integer(c_int64_t) :: ncol
        integer :: idim
        type(c_ptr) :: hist
        me%mdim = ndim
        ! the histogram comes back in work(1:ncol) when the caller's array can hold it
        hist = c_null_ptr
        ncol = 1
        do idim = 1, max(ndim,0)
            ncol = ncol*int(max(nodes(idim),1),c_int64_t)
        end do
        if (ndim >= 1 .and. xtrap /= 0.0_wp .and. int(nwrk,c_int64_t) >= ncol) hist = c_loc(work)
        rc = c_fit(int(ndim,c_int32_t), xdata, int(l1xdat,c_int32_t), ydata, wdata, &
                   int(ndata,c_int64_t), xmin, xmax, c_loc(nodes), xtrap, c_loc(coef), &
                   int(ncf,c_int64_t), int(nwrk,c_int64_t), hist, c_null_ptr)
        ierror = int(rc)
        if (rc > 0) then
            call report_fit(ierror)
        else if (rc < 0) then
            call report_library_failure(ierror,'splcc or splcw')
        end if
    end subroutine fit_common

    !> Spline value at one point; same arguments as the reference's splfe (:1258).
    function splfe(me,ndim,x,coef,xmin,xmax,nodes,ierror)
        class(splpak_type),intent(inout) :: me
        real(wp) :: splfe
        integer,intent(in) :: ndim
        real(wp),intent(in),target :: x(*)
        real(wp),intent(out),target :: coef(*)     ! intent as in the reference (:1264); only read
        real(wp),intent(in),target :: xmin(*), xmax(*)
        integer,intent(in),target :: nodes(*)
        integer,intent(out) :: ierror
        real(wp),target :: f(1)
        call eval_common(me,ndim,1_c_int64_t,c_loc(x),max(ndim,1),c_null_ptr,c_loc(coef),c_loc(xmin), &
                         c_loc(xmax),c_loc(nodes),c_loc(f),ierror)
        splfe = f(1)
    end function splfe

    !> Partial derivative at one point; same arguments as the reference's splde (:1089).
    function splde(me,ndim,x,nderiv,coef,xmin,xmax,nodes,ierror)
        class(splpak_type),intent(inout) :: me
        real(wp) :: splde
        integer,intent(in) :: ndim
        real(wp),intent(in),target :: x(*)
        integer,intent(in),target :: nderiv(*)
        real(wp),intent(out),target :: coef(*)
        real(wp),intent(in),target :: xmin(*), xmax(*)
        integer,intent(in),target :: nodes(*)
        integer,intent(out) :: ierror
        real(wp),target :: f(1)
        call eval_common(me,ndim,1_c_int64_t,c_loc(x),max(ndim,1),c_loc(nderiv),c_loc(coef),c_loc(xmin), &
                         c_loc(xmax),c_loc(nodes),c_loc(f),ierror)
        splde = f(1)
    end function splde

    !> Batch of values: x(ldx,nq) -> f(nq).  One kernel launch for all points.
    subroutine splfe_many(me,ndim,nq,x,ldx,coef,xmin,xmax,nodes,f,ierror)
        class(splpak_type),intent(inout) :: me
        integer,intent(in) :: ndim, nq, ldx
        real(wp),intent(in),target :: x(ldx,*)
        real(wp),intent(in),target :: coef(*)
        real(wp),intent(in),target :: xmin(*), xmax(*)
        integer,intent(in),target :: nodes(*)
        real(wp),intent(out),target :: f(*)
        integer,intent(out) :: ierror
        call eval_common(me,ndim,int(nq,c_int64_t),c_loc(x),ldx,c_null_ptr,c_loc(coef),c_loc(xmin), &
                         c_loc(xmax),c_loc(nodes),c_loc(f),ierror)
    end subroutine splfe_many

    !> Batch of partial derivatives (one `nderiv` pattern for all points).
    subroutine splde_many(me,ndim,nq,x,ldx,nderiv,coef,xmin,xmax,nodes,f,ierror)
        class(splpak_type),intent(inout) :: me
        integer,intent(in) :: ndim, nq, ldx
        real(wp),intent(in),target :: x(ldx,*)
        integer,intent(in),target :: nderiv(*)
        real(wp),intent(in),target :: coef(*)
        real(wp),intent(in),target :: xmin(*), xmax(*)
        integer,intent(in),target :: nodes(*)
        real(wp),intent(out),target :: f(*)
        integer,intent(out) :: ierror
        call eval_common(me,ndim,int(nq,c_int64_t),c_loc(x),ldx,c_loc(nderiv),c_loc(coef),c_loc(xmin), &
                         c_loc(xmax),c_loc(nodes),c_loc(f),ierror)
    end subroutine splde_many

    !> Value, gradient and (order = 2) Hessian of a batch of points in one pass: column i of
    !! f(ldf,nq) holds [ f, df/dx_1..df/dx_ndim, then for order 2 the upper triangle of the Hessian
    !! row by row ] at x(:,i) -- what 1 + ndim (+ ndim(ndim+1)/2) `evaluate` calls with the
    !! matching nderiv would return (splde, reference :1089-1240).
    subroutine splpak_derivs_many(me,ndim,nq,x,ldx,order,coef,xmin,xmax,nodes,f,ldf,ierror)
        class(splpak_type),intent(inout) :: me
        integer,intent(in) :: ndim, nq, ldx, order, ldf
        real(wp),intent(in),target :: x(ldx,*)
        real(wp),intent(in),target :: coef(*)
        real(wp),intent(in),target :: xmin(*), xmax(*)
        integer,intent(in),target :: nodes(*)
        real(wp),intent(out),target :: f(ldf,*)
        integer,intent(out) :: ierror
        integer(c_int32_t) :: rc
        me%mdim = ndim
        rc = c_eval_derivs(int(ndim,c_int32_t), int(nq,c_int64_t), c_loc(x), int(ldx,c_int32_t), &
                           int(order,c_int32_t), c_loc(coef), c_loc(xmin), c_loc(xmax), c_loc(nodes), &
                           c_loc(f), int(ldf,c_int32_t))
        ierror = int(rc)
        select case (ierror)
        case (0)
        case (101); call report(ierror,' splfe or splde - NDIM is less than 1')
        case (102); call report(ierror,' splfe or splde - NODES(IDIM) is less than  4for some IDIM')
        case (103); call report(ierror,' splfe or splde - XMIN(IDIM) = XMAX(IDIM) for some IDIM')
        case default
            if (ierror < 0) call report_library_failure(ierror,'evaluate_derivatives')
        end select
    end subroutine splpak_derivs_many

    subroutine eval_common(me,ndim,nq,x,ldx,nderiv,coef,xmin,xmax,nodes,f,ierror)
        class(splpak_type),intent(inout) :: me
        integer,intent(in) :: ndim, ldx
        integer(c_int64_t),intent(in) :: nq
        type(c_ptr),intent(in) :: x, nderiv, coef, xmin, xmax, nodes, f
        integer,intent(out) :: ierror
        integer(c_int32_t) :: rc
        me%mdim = ndim
        rc = c_eval(int(ndim,c_int32_t), nq, x, int(ldx,c_int32_t), nderiv, coef, xmin, xmax, nodes, f)
        ierror = int(rc)
        select case (ierror)
        case (0)
        case (101); call report(ierror,' splfe or splde - NDIM is less than 1')
        case (102); call report(ierror,' splfe or splde - NODES(IDIM) is less than  4for some IDIM')
        case (103); call report(ierror,' splfe or splde - XMIN(IDIM) = XMAX(IDIM) for some IDIM')
        case (104); call report(ierror,' splde - NDERIV(IDIM) IS less than 0 or greater than 2 for some IDIM')
        case default
            if (ierror < 0) call report_library_failure(ierror,'splfe or splde')
        end select
    end subroutine eval_common

end module splpak_module
