!> Re-creation of the reference's known-answer scenario (test/splpak_test_linear.f90:
!! 1-D, 10 nodes, 20 equispaced points of y = 2x, unit weights, xtrap = 1) against the
!! MI355X splpak_module.  Same assertions: fit error <= 1e-1, slope = 2 within 1e-12
!! (the reference re-tests the left slope for its "right" check, :85-89; here the right
!! slope is really tested).
program test_linear
    use splpak_module, wp => splpak_wp
    implicit none
    integer,parameter :: ndim = 1, nxdata = 20, nest = 100
    integer,dimension(ndim),parameter :: nodes = [10]
    integer,parameter :: ncol = product(nodes), nwrk = ncol*(ncol+1), ncf = ncol
    real(wp) :: xdata(ndim,nxdata), ydata(nxdata), wdata(nxdata), xmin(ndim), xmax(ndim)
    real(wp) :: work(nwrk), coef(ncf), x(ndim), xs(ndim,nest), fs(nest)
    real(wp) :: f, errmax, fleft, fright
    integer :: i, ierror
    integer(8) :: t0, t1, rate
    type(splpak_type) :: solver
    character(len=16) :: backend

    ! `<program> host`: the same scenario on the module's HOST solver (set_host; no GPU needed)
    call get_command_argument(1, backend)
    if (trim(backend) == 'host') call solver%set_host(.true.)

    xmin = 0.0_wp; xmax = 1.0_wp
    do i = 1, nxdata
        wdata(i) = 1.0_wp
        xdata(1,i) = real(i-1,wp)/real(nxdata-1,wp)
        ydata(i) = 2.0_wp*xdata(1,i)
    end do
    call solver%initialize(1,xdata,1,ydata,wdata,nxdata,xmin,xmax,nodes,1.0_wp,coef,ncf,work,nwrk,ierror)
    write(*,*) 'splcw ierror = ', ierror
    if (ierror /= 0) error stop 'error calling splcw'

    errmax = 0.0_wp
    call system_clock(t0, rate)
    do i = 1, nest
        x(1) = real(i-1,wp)/nest
        xs(1,i) = x(1)
        f = solver%evaluate(ndim,x,coef,xmin,xmax,nodes,ierror)
        if (ierror /= 0) error stop 'error calling splfe'
        errmax = max(errmax, abs(2.0_wp*x(1) - f))
    end do
    call system_clock(t1)
    ! scalar evaluate is a host computation (no kernel launch per point): the reference's 100-call loop
    write(*,'(A,ES12.4)') ' scalar evaluate loop seconds = ', real(t1-t0,8)/real(rate,8)
    write(*,*) 'splfe errmax [linear] = ', errmax
    if (errmax > 1.0e-1_wp) error stop 'errmax too large'

    fleft = solver%evaluate(ndim,[0.0_wp],[1],coef,xmin,xmax,nodes,ierror)
    if (ierror /= 0) error stop 'error calling splde'
    write(*,*) 'splde errmax left [linear] = ', fleft - 2.0_wp
    if (abs(fleft - 2.0_wp) > 1.0e-12_wp) error stop 'left slope wrong'
    fright = solver%evaluate(ndim,[1.0_wp],[1],coef,xmin,xmax,nodes,ierror)
    if (ierror /= 0) error stop 'error calling splde'
    write(*,*) 'splde errmax right [linear] = ', fright - 2.0_wp
    if (abs(fright - 2.0_wp) > 1.0e-12_wp) error stop 'right slope wrong'

    ! additive batched evaluation must agree with the scalar calls
    call solver%evaluate_many(ndim,nest,xs,ndim,coef,xmin,xmax,nodes,fs,ierror)
    if (ierror /= 0) error stop 'error calling evaluate_many'
    do i = 1, nest
        x(1) = xs(1,i)
        f = solver%evaluate(ndim,x,coef,xmin,xmax,nodes,ierror)
        ! host (scalar) and GPU (batched) evaluation: same formulas, different summation grouping
        if (abs(f - fs(i)) > 1.0e-14_wp*max(1.0_wp,abs(f))) error stop 'evaluate_many differs from evaluate'
    end do
    call solver%destroy()
    write(*,*) 'PASS test_linear'
end program test_linear
