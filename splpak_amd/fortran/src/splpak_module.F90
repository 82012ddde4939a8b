!> Drop-in `splpak_module` for MI355X: the public surface of jacobwilliams/splpak
!! (`splpak_wp`, `splpak_type` with the generics `initialize` => splcc/splcw,
!! `evaluate` => splfe/splde and `destroy`; reference src/splpak.F90:43-127) over the
!! HIP library `libsplpak_hip.so` (C ABI in include/splpak_hip.h).
!!
!! Argument lists, `ierror` codes and the text printed on `output_unit` follow the
!! reference (splcw :512-513, splcc :421-422, splde :1089, splfe :1258, cfaerr :399-407).
!! The fit and every batched evaluation run on the GPU and never fall back: without a GPU those calls
!! fail with a negative `ierror` and the library's message is printed.  A separate HOST solver
!! (splpak_host.F90, written from scratch for this package) runs only when the caller selects it
!! explicitly with `call solver%set_host(.true.)`, and in a -DREAL128 build (no GPU arithmetic).  The SCALAR
!! `evaluate` (one point per call, the reference's splfe/splde) is computed on the host by this
!! module itself (SURVEY 7.2 H6: one kernel launch per point would cost 10^4 x its arithmetic);
!! `evaluate_many` is the GPU path.
!!
!! Differences a caller can observe (documented in INTEGRATION.md):
!!  * `work` is not used as scratch.  The reference's 106 check on `nwrk` is kept; the
!!    larger requirement of the dense solver (error 107 via suprls 32, :1443-1454) is
!!    not enforced, because the GPU path does not need the ncol*(ncol+1) array (550 GB
!!    at 64^3 nodes).  After a successful fit with xtrap /= 0 `work(1:ncol)` holds the
!!    sparse-area histogram exactly as the reference leaves it (:879-907).
!!  * additive: `evaluate_many` evaluates a batch of points in one kernel launch,
!!    `evaluate_derivatives` value + gradient (+ Hessian) of a batch in one pass,
!!    `last_fit_info` returns the diagnostics of the last fit (the residual norm `reserr` that the
!!    reference computes, suprls :1693, and drops, splcw :690; row counts; refinement steps).
!!
!! Build with -DREAL32 for single precision storage (as the reference, :33-41); -DREAL128 builds the
!! module on the host solver alone (quad precision has no MI355X arithmetic).
module splpak_module

    use iso_c_binding
    use iso_fortran_env, only: real32, real64, real128, output_unit
    use splpak_host_solver, only: hk, host_basis, host_fit

    implicit none

    private

#ifdef REAL32
    integer,parameter :: wp = real32
#elif REAL128
    integer,parameter :: wp = real128
#else
    integer,parameter :: wp = real64
#endif

    integer,parameter,public :: splpak_wp = wp   !! working precision

    integer,save :: no_gpu_optin = -1            !! SPLPAK_HOST_IF_NO_GPU as read once (-1: not read yet)
    logical,save :: no_gpu_said = .false.        !! the one line about it has been printed

    type,public :: splpak_type
        private
        integer :: mdim = 0    !! dimension of the last call (the reference keeps scratch here, :95-111)
        real(real64) :: info(10) = 0.0_real64   !! diagnostics of the last fit (include/splpak_hip.h, `info`)
        integer :: ngpus = 1   !! GPUs of this node the fit is spread over (set_gpus)
#ifdef REAL128
        logical :: host = .true.    !! quad precision: the host solver is the only path
#else
        logical :: host = .false.   !! .true. after set_host(.true.): fits and batches run on the host solver (never a fallback)
#endif
    contains
        private
        generic,public   :: initialize    => splcc, splcw        !! fit
        generic,public   :: evaluate      => splfe, splde        !! one point
        generic,public   :: evaluate_many => splfe_many, splde_many  !! batch of points (additive)
        procedure,public :: evaluate_derivatives => splpak_derivs_many !! value + gradient (+ Hessian) of a batch (additive)
        procedure,public :: destroy       => destroy_splpak
        procedure,public :: last_fit_info => splpak_last_fit_info   !! reserr, row counts, ... of the last fit (additive)
        procedure,public,nopass :: set_option => splpak_set_option    !! a named option of the HIP library for the following fits (additive)
        procedure,public :: set_gpus      => splpak_set_gpus        !! spread the following fits over n GPUs of this node (additive)
        procedure,public :: set_host      => splpak_set_host        !! run the following calls on the host solver (additive; explicit, never a fallback)
        procedure,private :: splcc
        procedure,private :: splcw
        procedure,private :: splfe
        procedure,private :: splde
        procedure,private :: splfe_many
        procedure,private :: splde_many
        procedure,private :: splpak_derivs_many
    end type splpak_type

#ifndef REAL128
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
#ifndef REAL32
        integer(c_int32_t) function c_fit_multi(ngpus,ndim,xdata,l1xdat,ydata,wdata,ndata,xmin,xmax,nodes,xtrap,&
                                                coef,ncf,nwrk,hist,info) bind(C,name='splpak_fit_multi_f64')
            import :: c_int32_t, c_int64_t, c_ptr, wp
            integer(c_int32_t),value :: ngpus, ndim, l1xdat
            integer(c_int64_t),value :: ndata, ncf, nwrk
            type(c_ptr),value :: xdata, ydata, wdata, xmin, xmax, nodes, coef, hist, info
            real(wp),value :: xtrap
        end function c_fit_multi
#endif
        subroutine c_shutdown() bind(C,name='splpak_shutdown')
        end subroutine c_shutdown
        integer(c_int32_t) function c_set_default_option(name,val) bind(C,name='splpak_set_default_option')
            import :: c_int32_t, c_char
            character(kind=c_char) :: name(*), val(*)
        end function c_set_default_option
        integer(c_int32_t) function c_last_error(buf,buflen) bind(C,name='splpak_last_error_message')
            import :: c_int32_t, c_char
            character(kind=c_char) :: buf(*)
            integer(c_int32_t),value :: buflen
        end function c_last_error
    end interface
#endif

    contains

    !> Release the object's state (the reference reallocates scratch here, :136-165).  Called
    !! without `ndim` it also releases the device memory the library keeps between fits of the same
    !! grid (band factor storage, staging buffers); the next fit simply allocates again.
    subroutine destroy_splpak(me,ndim)
        class(splpak_type),intent(inout) :: me
        integer,intent(in),optional :: ndim
        me%mdim = 0
        me%info = 0.0_real64
        if (present(ndim)) then
            me%mdim = ndim
#ifndef REAL128
        else
            call c_shutdown()
#endif
        end if
    end subroutine destroy_splpak

    !> `call solver%set_host(.true.)`: the following `initialize` / `evaluate_many` / `evaluate_derivatives` calls of this
    !! object run on the HOST solver (splpak_host.F90: banded normal equations + Cholesky + refinement against the rows,
    !! real64 arithmetic) instead of the MI355X -- for machines without a GPU and for small problems.  Explicit only: the
    !! GPU path never falls back to it.  `.false.` returns to the GPU (ignored in a -DREAL128 build, which has no GPU path).
    subroutine splpak_set_host(me,flag)
        class(splpak_type),intent(inout) :: me
        logical,intent(in) :: flag
#ifdef REAL128
        me%host = .true.
#else
        me%host = flag
#endif
    end subroutine splpak_set_host

    !> `call solver%set_option('solver','pcg+direct',ierror)`: a named option of the HIP library (include/splpak_hip.h,
    !! splpak_set_default_option; INTEGRATION.md lists them) for the fits that follow -- process wide, what SPLPAK_<NAME> in the
    !! environment would set.  ierror = 0, or -3 for an unknown name (the library's message is printed).  No effect on the host solver.
    subroutine splpak_set_option(name,value,ierror)
        character(len=*),intent(in) :: name, value
        integer,intent(out) :: ierror
        ierror = 0
#ifndef REAL128
        ierror = int(c_set_default_option(trim(name)//c_null_char, trim(value)//c_null_char))
        if (ierror < 0) call report_library_failure(ierror,'set_option')
#endif
    end subroutine splpak_set_option

    !> The following `initialize` calls of this object use `n` GPUs of the node: the points are sharded
    !! and the band of the normal equations is distributed over them (include/splpak_hip.h,
    !! splpak_fit_multi_f64); n <= 1 is the single-GPU fit.  Same arguments, same results.
    subroutine splpak_set_gpus(me,n)
        class(splpak_type),intent(inout) :: me
        integer,intent(in) :: n
        me%ngpus = max(n,1)
    end subroutine splpak_set_gpus

    !> Diagnostics of the last `initialize` of this object.  `reserr` is the residual norm
    !! ||rows*coef - rhs||_2 over data and constraint rows that the reference computes in suprls
    !! (:1693) and drops in splcw (:690, :1052).
    subroutine splpak_last_fit_info(me,reserr,ndata_rows,nconstraint_rows,refine_steps,optimality,on_host)
        class(splpak_type),intent(in) :: me
        real(wp),intent(out),optional :: reserr        !! residual norm of the fitted system
        integer,intent(out),optional :: ndata_rows       !! data rows used (non-zero weight)
        integer,intent(out),optional :: nconstraint_rows !! derivative-constraint rows of data-sparse nodes (:921-1046)
        integer,intent(out),optional :: refine_steps     !! iterative-refinement steps taken
        real(wp),intent(out),optional :: optimality      !! componentwise backward error of coef w.r.t. the rows
        logical,intent(out),optional :: on_host          !! .true. if this object's fits run on the host solver (set_host)
        if (present(on_host)) on_host = me%host
        if (present(reserr)) reserr = real(me%info(9),wp)
        if (present(ndata_rows)) ndata_rows = int(me%info(1))
        if (present(nconstraint_rows)) nconstraint_rows = int(me%info(2))
        if (present(refine_steps)) refine_steps = int(me%info(3))
        if (present(optimality)) optimality = real(me%info(10),wp)
    end subroutine splpak_last_fit_info

    !> ` IERR=nnnnn` + message on output_unit, as the reference's cfaerr (:399-407).
    subroutine report(ierr,mess)
        integer,intent(in) :: ierr
        character(len=*),intent(in) :: mess
        if (ierr /= 0) write (output_unit,'(A,I5)') ' IERR=', ierr
        write (output_unit,'(A)') trim(mess)
    end subroutine report

    !> a negative status is an infrastructure failure (no GPU, out of device memory, ...)
    !> .true. when a call has to run on the host solver whatever the GPU could do: `set_host(.true.)`, or more than four
    !! dimensions -- the reference takes any ndim >= 1 (:716-722, :1166-1172; "1..4" is a remark in its documentation, :1099),
    !! the HIP kernels are written for 1..4 (include/splpak_hip.h SPLPAK_MAXDIM), the host solver is written for any ndim.
    pure logical function host_takes(me,ndim)
        class(splpak_type),intent(in) :: me
        integer,intent(in) :: ndim
        host_takes = me%host .or. ndim > 4
    end function host_takes

    !> Opt-in for machines without a GPU (fpm dependents running their tests, SURVEY section 8f-4): with
    !! SPLPAK_HOST_IF_NO_GPU=1 in the environment a call that the HIP library refuses with "no usable device" (-1) runs on
    !! the host solver instead, and says so once per process on output_unit.  Never a default, never silent; any other
    !! failure of the library is reported as before.
    logical function host_if_no_gpu(rc)
        integer,intent(in) :: rc
        character(len=8) :: val
        integer :: n, stat
        host_if_no_gpu = .false.
        if (rc /= -1) return
        if (no_gpu_optin < 0) then
            no_gpu_optin = 0
            call get_environment_variable('SPLPAK_HOST_IF_NO_GPU', val, n, stat)
            if (stat == 0 .and. n >= 1) then
                if (val(1:1) == '1') no_gpu_optin = 1
            end if
        end if
        host_if_no_gpu = no_gpu_optin == 1
        if (host_if_no_gpu .and. .not. no_gpu_said) then
            no_gpu_said = .true.
            write(output_unit,'(a)') ' splpak: no usable GPU (HIP library status -1); '// &
                                     'SPLPAK_HOST_IF_NO_GPU=1: running on the host solver'
        end if
    end function host_if_no_gpu

    subroutine report_library_failure(ierr,who)
        integer,intent(in) :: ierr
        character(len=*),intent(in) :: who
        character(kind=c_char) :: buf(512)
        character(len=512) :: msg
        integer :: n, i
        n = 0
        buf = ' '
#ifndef REAL128
        n = c_last_error(buf, 512_c_int32_t)
#endif
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
        class(splpak_type),intent(inout),target :: me
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
        if (host_takes(me,ndim)) then
            call fit_on_host(me,ndim,xdata,l1xdat,ydata,wdata,ndata,xmin,xmax,nodes,xtrap,coef,ncf,work,nwrk,ierror)
            return
        end if
#ifndef REAL128
#ifndef REAL32
        if (me%ngpus > 1) then
            rc = c_fit_multi(int(me%ngpus,c_int32_t), int(ndim,c_int32_t), xdata, int(l1xdat,c_int32_t), ydata, wdata, &
                             int(ndata,c_int64_t), xmin, xmax, c_loc(nodes), xtrap, c_loc(coef), &
                             int(ncf,c_int64_t), int(nwrk,c_int64_t), hist, c_loc(me%info))
        else
#endif
        rc = c_fit(int(ndim,c_int32_t), xdata, int(l1xdat,c_int32_t), ydata, wdata, &
                   int(ndata,c_int64_t), xmin, xmax, c_loc(nodes), xtrap, c_loc(coef), &
                   int(ncf,c_int64_t), int(nwrk,c_int64_t), hist, c_loc(me%info))
#ifndef REAL32
        end if
#endif
        ierror = int(rc)
        if (host_if_no_gpu(ierror)) then
            call fit_on_host(me,ndim,xdata,l1xdat,ydata,wdata,ndata,xmin,xmax,nodes,xtrap,coef,ncf,work,nwrk,ierror)
            return
        end if
        if (rc > 0) then
            call report_fit(ierror)
        else if (rc < 0) then
            call report_library_failure(ierror,'splcc or splcw')
        end if
#endif
    end subroutine fit_common

    !> The fit on the host solver (set_host / REAL128): the reference's argument checks in its order (:716-781), then
    !! splpak_host_solver's banded normal equations.  Same `ierror` codes and messages as the GPU path.
    subroutine fit_on_host(me,ndim,xdata,l1xdat,ydata,wdata,ndata,xmin,xmax,nodes,xtrap,coef,ncf,work,nwrk,ierror)
        class(splpak_type),intent(inout) :: me
        integer,intent(in) :: ndim, l1xdat, ncf, nwrk, ndata
        type(c_ptr),intent(in) :: xdata, ydata, wdata, xmin, xmax
        integer,intent(in) :: nodes(*)
        real(wp),intent(in) :: xtrap
        real(wp) :: coef(*), work(*)
        integer,intent(out) :: ierror
        real(wp),pointer :: px(:,:), py(:), pw(:), pmin(:), pmax(:)
        real(wp),target :: wdummy(1)
        real(wp),allocatable :: hst(:)
        integer :: idim, nwrk1
        integer(c_int64_t) :: ncol
        logical :: weighted, want
        ierror = 0
        if (ndim < 1) ierror = 101
        if (ierror == 0) then
            call c_f_pointer(xmin, pmin, [ndim])
            call c_f_pointer(xmax, pmax, [ndim])
            ncol = 1
            do idim = 1, ndim
                if (nodes(idim) < 4) then
                    ierror = 102
                    exit
                end if
                if (pmax(idim) - pmin(idim) == 0.0_wp) then
                    ierror = 103
                    exit
                end if
                ncol = ncol*int(nodes(idim),c_int64_t)
            end do
        end if
        if (ierror == 0) then
            if (ncol > int(ncf,c_int64_t)) ierror = 104
        end if
        if (ierror == 0 .and. ndata < 1) ierror = 105
        if (ierror == 0) then
            nwrk1 = 1
            if (xtrap /= 0.0_wp) nwrk1 = int(min(ncol+1, int(huge(1),c_int64_t)))
            if (nwrk - nwrk1 + 1 < 1) ierror = 106
        end if
        if (ierror /= 0) then
            call report_fit(ierror)
            return
        end if
        call c_f_pointer(xdata, px, [l1xdat, ndata])
        call c_f_pointer(ydata, py, [ndata])
        weighted = c_associated(wdata)
        if (weighted) then
            call c_f_pointer(wdata, pw, [ndata])
        else
            wdummy = -1.0_wp
            pw => wdummy
        end if
        want = xtrap /= 0.0_wp .and. int(nwrk,c_int64_t) >= ncol
        allocate(hst(int(ncol)))
        call host_fit(ndim, px, l1xdat, py, pw, weighted, ndata, pmin, pmax, nodes, xtrap, coef, hst, want, me%info, ierror)
        if (want) work(1:int(ncol)) = hst
        if (ierror /= 0) call report_fit(ierror)
    end subroutine fit_on_host

    !> Spline value at one point; same arguments as the reference's splfe (:1258).  Host computation.
    function splfe(me,ndim,x,coef,xmin,xmax,nodes,ierror)
        class(splpak_type),intent(inout) :: me
        real(wp) :: splfe
        integer,intent(in) :: ndim
        real(wp),intent(in) :: x(*)
        real(wp),intent(out) :: coef(*)            ! intent as in the reference (:1264); only read
        real(wp),intent(in) :: xmin(*), xmax(*)
        integer,intent(in) :: nodes(*)
        integer,intent(out) :: ierror
        integer :: nderiv(max(ndim,1))
        nderiv = 0
        splfe = eval_point(me,ndim,x,nderiv,coef,xmin,xmax,nodes,ierror)
    end function splfe

    !> Partial derivative at one point; same arguments as the reference's splde (:1089).  Host computation.
    function splde(me,ndim,x,nderiv,coef,xmin,xmax,nodes,ierror)
        class(splpak_type),intent(inout) :: me
        real(wp) :: splde
        integer,intent(in) :: ndim
        real(wp),intent(in) :: x(*)
        integer,intent(in) :: nderiv(*)
        real(wp),intent(out) :: coef(*)
        real(wp),intent(in) :: xmin(*), xmax(*)
        integer,intent(in) :: nodes(*)
        integer,intent(out) :: ierror
        splde = eval_point(me,ndim,x,nderiv,coef,xmin,xmax,nodes,ierror)
    end function splde

    !> The scalar evaluation: checks and `ierror` as splde (:1166-1194: 101/102/103 return 0, 104 is
    !! reported and the value is still computed), then the separable form of the reference's window
    !! sum (:1197-1236): per dimension the (at most) four 1-D factors of the window
    !! [ibmn, ibmx] (:1201-1209) are tabulated once, and the 4^ndim products are accumulated with the
    !! first dimension running fastest.  Any ndim >= 1 (the reference documents 1..4, :1099).
    function eval_point(me,ndim,x,nderiv,coef,xmin,xmax,nodes,ierror) result(f)
        class(splpak_type),intent(inout) :: me
        integer,intent(in) :: ndim
        real(wp),intent(in) :: x(*)
        integer,intent(in) :: nderiv(*)
        real(wp),intent(in) :: coef(*)
        real(wp),intent(in) :: xmin(*), xmax(*)
        integer,intent(in) :: nodes(*)
        integer,intent(out) :: ierror
        real(wp) :: f
        real(hk) :: tab(4,max(ndim,1)), dx, s, t, prod, acc
        integer :: first(max(ndim,1)), width(max(ndim,1)), k(max(ndim,1)), stride(max(ndim,1))
        integer :: idim, it, ibmn, ibmx, j, icol, ider
        f = 0.0_wp
        ierror = 0
        me%mdim = ndim
        if (ndim < 1) then
            ierror = 101
            call report(ierror,' splfe or splde - NDIM is less than 1')
            return
        end if
        do idim = 1, ndim
            if (nodes(idim) < 4) then
                ierror = 102
                call report(ierror,' splfe or splde - NODES(IDIM) is less than  4for some IDIM')
                return
            end if
            if (xmax(idim) - xmin(idim) == 0.0_wp) then
                ierror = 103
                call report(ierror,' splfe or splde - XMIN(IDIM) = XMAX(IDIM) for some IDIM')
                return
            end if
        end do
        do idim = 1, ndim
            if (nderiv(idim) < 0 .or. nderiv(idim) > 2) then
                ierror = 104
                call report(ierror,' splde - NDERIV(IDIM) IS less than 0 or greater than 2 for some IDIM')
                exit                                        ! the reference reports and computes on (:1190-1194)
            end if
        end do
        icol = 1
        do idim = 1, ndim
            stride(idim) = icol
            icol = icol*nodes(idim)
            dx = (real(xmax(idim),hk) - real(xmin(idim),hk))/real(nodes(idim)-1,hk)
            s = 1.0_hk/dx
            t = s*(real(x(idim),hk) - real(xmin(idim),hk))
            it = int(max(min(t,2.0e9_hk),-2.0e9_hk))     ! truncation toward zero, saturating
            ibmn = min(max(it-1,0),nodes(idim)-2)
            ibmx = max(min(it+2,nodes(idim)-1),1)
            first(idim) = ibmn
            width(idim) = ibmx - ibmn + 1
            ider = min(max(nderiv(idim),0),2)
            do j = 1, width(idim)
                tab(j,idim) = host_basis(ibmn+j-1, nodes(idim), ider, real(x(idim),hk), &
                                         real(xmin(idim),hk) + real(ibmn+j-1,hk)*dx, s)
            end do
        end do
        acc = 0.0_hk
        k = 1
        do
            prod = 1.0_hk
            icol = 1
            do idim = 1, ndim
                prod = prod*tab(k(idim),idim)
                icol = icol + (first(idim) + k(idim) - 1)*stride(idim)
            end do
            acc = acc + real(coef(icol),hk)*prod
            idim = 1                                        ! odometer, first dimension fastest (:1228-1232)
            do while (idim <= ndim)
                k(idim) = k(idim) + 1
                if (k(idim) <= width(idim)) exit
                k(idim) = 1
                idim = idim + 1
            end do
            if (idim > ndim) exit
        end do
        f = real(acc,wp)
    end function eval_point

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
        integer :: iq, idm, jdm, col, nder(max(ndim,1)), ie
        me%mdim = ndim
        rc = 0
#ifndef REAL128
        if (.not. host_takes(me,ndim)) then
            rc = c_eval_derivs(int(ndim,c_int32_t), int(nq,c_int64_t), c_loc(x), int(ldx,c_int32_t), &
                               int(order,c_int32_t), c_loc(coef), c_loc(xmin), c_loc(xmax), c_loc(nodes), &
                               c_loc(f), int(ldf,c_int32_t))
        end if
#endif
        if (host_takes(me,ndim) .or. host_if_no_gpu(int(rc))) then      ! host solver: every column is the scalar splde of its pattern
            ierror = 0
            if (order < 1 .or. order > 2 .or. ldf < 1 + ndim + merge(ndim*(ndim+1)/2, 0, order == 2)) then
                ierror = -3
                call report(ierror,' evaluate_derivatives - order must be 1 or 2 and ldf large enough')
                return
            end if
            do iq = 1, nq
                nder = 0
                f(1,iq) = eval_point(me,ndim,x(:,iq),nder,coef,xmin,xmax,nodes,ie)
                if (ie /= 0) then
                    ierror = ie
                    return
                end if
                do idm = 1, ndim
                    nder = 0
                    nder(idm) = 1
                    f(1+idm,iq) = eval_point(me,ndim,x(:,iq),nder,coef,xmin,xmax,nodes,ie)
                end do
                if (order == 2) then
                    col = 1 + ndim
                    do idm = 1, ndim
                        do jdm = idm, ndim
                            nder = 0
                            nder(idm) = nder(idm) + 1
                            nder(jdm) = nder(jdm) + 1
                            col = col + 1
                            f(col,iq) = eval_point(me,ndim,x(:,iq),nder,coef,xmin,xmax,nodes,ie)
                        end do
                    end do
                end if
            end do
            return
        end if
#ifdef REAL128
        rc = -1
#endif
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
        real(wp),pointer :: px(:,:), pc(:), pmin(:), pmax(:), pf(:)
        integer,pointer :: pn(:), pd(:)
        integer :: iq, ie, zero(max(ndim,1)), ncol, idim
        me%mdim = ndim
        rc = 0
#ifndef REAL128
        if (.not. (host_takes(me,ndim) .and. ndim >= 1)) &
            rc = c_eval(int(ndim,c_int32_t), nq, x, int(ldx,c_int32_t), nderiv, coef, xmin, xmax, nodes, f)
#endif
        if ((host_takes(me,ndim) .or. host_if_no_gpu(int(rc))) .and. ndim >= 1) then   ! host solver: a loop over the scalar evaluation
            call c_f_pointer(nodes, pn, [ndim])
            ncol = 1
            do idim = 1, ndim
                ncol = ncol*max(pn(idim),1)
            end do
            call c_f_pointer(x, px, [ldx, int(nq)])
            call c_f_pointer(coef, pc, [ncol])
            call c_f_pointer(xmin, pmin, [ndim])
            call c_f_pointer(xmax, pmax, [ndim])
            call c_f_pointer(f, pf, [int(nq)])
            zero = 0
            ierror = 0
            do iq = 1, int(nq)
                if (c_associated(nderiv)) then
                    call c_f_pointer(nderiv, pd, [ndim])
                    pf(iq) = eval_point(me,ndim,px(:,iq),pd,pc,pmin,pmax,pn,ie)
                else
                    pf(iq) = eval_point(me,ndim,px(:,iq),zero,pc,pmin,pmax,pn,ie)
                end if
                if (ie /= 0) ierror = ie
                if (ie /= 0 .and. ie /= 104) return
            end do
            return
        end if
#ifdef REAL128
        rc = -1
#endif
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
